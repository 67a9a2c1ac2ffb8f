"""Sample the GPU's shader clock, power and temperature from sysfs while a command runs (no rocm-smi process per sample).

    python tools/gpu_clock_sampler.py <out.csv> <period_ms> -- <command ...>

One CSV row per period: seconds since start, sclk MHz (the starred level of pp_dpm_sclk, or hwmon freq1_input), socket
power in W (hwmon power1_average / power1_input), temperatures in C.  The command's own stdout/stderr pass through.
Whatever the box does not expose is left empty; a `rocm-smi` snapshot before and after is written next to the CSV."""
import glob, os, subprocess, sys, threading, time


def our_card():
    """sysfs node of the GPU THIS job runs on.  A box of the pool is one GPU of an 8-GPU host: sysfs shows all eight, the
    device cgroup lets us open only our own render node.  GPU_PCI_BUS_ID (e.g. 0000:75:00.0) overrides."""
    bus = os.environ.get("GPU_PCI_BUS_ID", "").lower()
    if bus and os.path.exists(f"/sys/bus/pci/devices/{bus}/pp_dpm_sclk"):
        return f"/sys/bus/pci/devices/{bus}"
    candidates = []
    for node in sorted(glob.glob("/sys/class/drm/renderD*")):
        dev = os.path.join(node, "device")
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002" or not os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                continue
            candidates.append(dev)
            fd = os.open("/dev/dri/" + os.path.basename(node), os.O_RDWR)
            os.close(fd)
            return os.path.realpath(dev)
        except OSError:
            continue
    return os.path.realpath(candidates[0]) if candidates else None


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return ""


def starred_mhz(text):
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            for tok in line.replace("Mhz", " ").replace("MHz", " ").split():
                try:
                    return float(tok)
                except ValueError:
                    continue
    return ""


def main():
    out, period = sys.argv[1], float(sys.argv[2]) * 1e-3
    cmd = sys.argv[sys.argv.index("--") + 1 :]
    dev = our_card()
    hw = (glob.glob(os.path.join(dev, "hwmon", "hwmon*")) or [None])[0] if dev else None
    stop = threading.Event()
    rows = []

    def snapshot(tag):
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel"], capture_output=True, text=True, timeout=30)
            open(out + f".{tag}.smi.txt", "w").write(r.stdout + r.stderr)
        except Exception as e:  # noqa: BLE001 - diagnostics only
            open(out + f".{tag}.smi.txt", "w").write(repr(e))

    def loop():
        t0 = time.perf_counter()
        while not stop.is_set():
            sclk = starred_mhz(read(os.path.join(dev, "pp_dpm_sclk"))) if dev else ""
            mclk = starred_mhz(read(os.path.join(dev, "pp_dpm_mclk"))) if dev else ""
            power = freq = temps = ""
            if hw:
                p = read(os.path.join(hw, "power1_average")) or read(os.path.join(hw, "power1_input"))
                power = float(p) / 1e6 if p else ""
                f = read(os.path.join(hw, "freq1_input"))
                freq = float(f) / 1e6 if f else ""
                temps = "/".join(str(int(read(t)) // 1000) for t in sorted(glob.glob(os.path.join(hw, "temp*_input"))) if read(t))
            rows.append((time.perf_counter() - t0, sclk, freq, mclk, power, temps))
            stop.wait(period)

    snapshot("before")
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.run(cmd).returncode
    stop.set()
    th.join()
    snapshot("after")
    with open(out, "w") as f:
        f.write(f"# device {dev} hwmon {hw}; command: {' '.join(cmd)}\n")
        f.write("seconds,sclk_mhz_pp_dpm,sclk_mhz_hwmon,mclk_mhz,power_w,temps_c\n")
        for r in rows:
            f.write(",".join(f"{x:.3f}" if isinstance(x, float) else str(x) for x in r) + "\n")
    sys.exit(rc)


if __name__ == "__main__":
    main()
