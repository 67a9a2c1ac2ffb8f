"""The default bench command N times (fresh process each), with RT_HIP_DEBUG_FRAME=1: per run the wall time per step, the time
between the kernel's end and the frame being in the caller's buffer, and what the carrier reported for the timed frames —
how many of the 127 bands were delivered before the stream's drain was observed, and how long the rest took."""
import json
import os
import re
import subprocess
import sys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
for i in range(n):
    out = subprocess.run([sys.executable, "bench.py", "--cpu-baseline-seconds", "0", "--no-kernel-only"], capture_output=True, text=True, env=dict(os.environ, RT_HIP_DEBUG_FRAME="1"))
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    early = [int(x) for x in re.findall(r"delivered: (\d+) of", out.stderr)][-20:]
    rest = [float(x) for x in re.findall(r"the rest took ([0-9.]+) us", out.stderr)][-20:]
    print(f"run {i:2d}: ms_per_step {line['ms_per_step']:.4f} kernel {line['roofline']['kernel_ms']:.4f} after_kernel {line['drop_in_breakdown']['after_kernel_ms']:.4f} | timed frames: bands early min {min(early)} of 127, rest max {max(rest):.1f} us", flush=True)
