#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (= W*H*spp / wall-seconds / 1e6) and wall-clock for a 1920x1080x256spp render of
scenes/basic.toml (BASELINE.json `metric`), on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # one process per GPU (what the driver launches)
    python bench.py --gpus N --steps K --warmup W            # not under torchrun: ONE process drives all N GPUs through
                                                            # rt_hip_create_multi — the form the reference's blocking
                                                            # render() can use (--same-device rehearses it on one GPU)

A "step" is one whole `render(scene, back_buffer)` as the reference makes it: host scene
columns in, finished frame in the caller's HOST buffer out, ONE frame at a time — scene fingerprint check (the columns
were uploaded during warm-up and are not re-sent while unchanged), kernel(s), for N > 1 the RCCL gather to rank 0 and the
de-interleave, and the delivery of the frame into the caller's (pageable) back buffer.  `value` is that drop-in rate; the
kernel-only rate (scene resident, frame left in HBM) is carried beside it as `kernel_only`.  The frame is fixed as N grows:
scaling is STRONG.

N > 1 under torchrun (round 4).  The step exists in three forms:
  * `library`      — THE ONE north_star NAMES, and `value` whenever it came up: every process is one rank of the module's
                     gathering renderer (rt_hip_create + rt_hip_join_ranks: ncclCommInitRank; ONE ncclGather over xGMI
                     inside librt_hip.so; rank 0 assembles into its frame and delivers it to its caller's back buffer);
  * `shared_frame` — side key: every process is one rank of a FRAME GROUP (rt_hip_join_frame_group): rank 0's back buffer
                     is a shared mapping every rank process maps and page-locks, every rank's kernel stores its stripes
                     straight into it over its own PCIe link — no data-path collective, no RCCL;
  * `torch`        — side key: torch.distributed.gather + rt_hip_assemble_device + a copy to a pinned host frame: building
                     blocks that every ROCm installation exercises.
Each form runs in a CHILD PROCESS OF ITS OWN per rank, started by this (torchrun-launched) process BEFORE anything has
touched a GPU: one form = one context and at most one communicator per GPU, all of it gone when the child exits; a form that
does not come back within its deadline is killed (the exact child, by pid) and reported as hung, the remaining forms still
run, and the benchmark then EXITS NON-ZERO after printing its line.  The torchrun-launched processes themselves only keep
a gloo group (ports, names, verdicts); they never initialise a GPU.  Inside a child the ranks vote before anything
collective is entered and the joins have deadlines of their own; the first frame of every form must equal, bit for bit,
the frame rank 0 renders alone on its GPU.  Every N > 1 line carries what the transport reports about itself (`rccl`: for
the library form ncclCommCount / ncclCommUserRank / ncclCommCuDevice of every rank's communicator), every rank's own kernel
time (`per_rank`), rank 0's split of a step (`drop_in_breakdown`) and all forms' figures and verdicts (`paths`).

The `roofline` object prices the render kernel against the FP32 vector-ALU peak — the bound SURVEY.md §8d identifies for
this path (a 3-sphere scene is ~100 bytes; the only compulsory HBM traffic is the 4 B/pixel frame) — from the kernel's
duration measured with HIP events on its launch stream (recorded by the module around every launch of the timed region
and read back through rt_hip_stats), carries the HBM figures next to it, and — `roofline.issue` — how close the kernel's
own instruction stream runs to the vector issue rate (counter figures of profiles/pmc_counters.json, used only when they
were measured on the kernels of this build).  `cpu_baseline` times the reference-faithful CPU model (oracle/, mt19937
mode) on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
ROOT_CGROUP = Path("/sys/fs/cgroup")
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's host driver only supports dmabuf IPC (RCCL across processes)

FP32_VALU_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (vector)"
HBM_PEAK_GBPS = 8000.0  # same guide, HBM3E spec
SHADER_CLOCK_HZ = 2.4e9  # the clock the FP32 peak above is quoted at (under load the card holds 2.36-2.40 GHz: profiles/r03/config5_streamed/)


def algorithmic_flops(samples: int, segments: int, n_spheres: int, n_planes: int) -> float:
    """S*70 + segments*(22*N_s + 16*N_p + 60)."""
    return samples * 70.0 + segments * (22.0 * n_spheres + 16.0 * n_planes + 60.0)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_quota() -> float | None:
    """CPUs' worth of time this process's cgroup grants per period (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us); None = no limit."""
    try:
        quota, period = (ROOT_CGROUP / "cpu.max").read_text().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = float((ROOT_CGROUP / "cpu" / "cpu.cfs_quota_us").read_text())
        period = float((ROOT_CGROUP / "cpu" / "cpu.cfs_period_us").read_text())
        return quota / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def cpu_baseline(scene_name: str, width: int, height: int, target_seconds: float) -> dict:
    """Reference-faithful CPU model (oracle, mt19937 mode) on all host cores this process may use, on a bounded
    sample: the same scene and frame size at a reduced spp chosen to take about `target_seconds`."""
    import rt_amd
    from oracle import binding as oracle  # cpu_baseline leg: the oracle is the thing timed here, by design

    allowed = len(os.sched_getaffinity(0))
    quota = cpu_quota()  # a container's CPU-time limit: threads beyond it only take turns
    cores = max(1, min(allowed, math.ceil(quota))) if quota else allowed
    scene = rt_amd.Scene.named(scene_name)
    spp = 1
    while True:  # grow the sample until it runs for at least half the target (thread start-up skews tiny probes)
        scene.set_sampling(spp)
        _, _, stats = oracle.render_mt19937(scene.describe(width, height), width, height, threads=cores)
        if stats["seconds"] >= 0.5 * target_seconds or spp >= 256:
            break
        rate = stats["primary_samples"] / max(stats["seconds"], 1e-6)
        spp = int(max(spp + 1, min(256, target_seconds * rate / (width * height))))
    return {
        "value": round(stats["primary_samples"] / stats["seconds"] / 1e6, 3),
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{scene_name} {width}x{height} at {spp} spp (of the workload's spp), {stats['seconds']:.1f} s on {cores} threads of {cpu_model()} "
        + (f"(the process may run on {allowed} CPUs but its cgroup grants the CPU time of {quota:g}: one thread per granted CPU); " if quota and cores < allowed else "(every CPU the process may run on); ")
        + "CPU restatement of mg_ray_tracer (thread_local mt19937, recursion, AoS scan), g++ -O3 -mavx2 -mfma -ffast-math -ffp-contract=fast "
        "(-march=native is not used: the .so is built on another host)",
    }


PHASE_KEYS = ("render_ms", "gather_ms", "assemble_ms", "copy_ms", "host_issue_ms", "host_wait_ms")
TILT = ((0.2, 1.2, 3.0), (0.0, -0.15, -1.0))  # --tilt / the `interactive` leg: look down by 8.5 degrees from beside the scene's own eye point


def process_cpu_seconds() -> float:
    """CPU time of this process so far, all its threads (the module's carrier helpers included): user + system."""
    import resource

    usage = resource.getrusage(resource.RUSAGE_SELF)
    return usage.ru_utime + usage.ru_stime


def cgroup_throttle() -> dict | None:
    """This process's cgroup CPU-bandwidth record (cgroup v2 cpu.stat): periods in which its threads were stopped because the
    group had used up its quota, and for how long.  A throttled group's threads ALL stop — the module's carrier helpers with
    them, whose work the caller's thread then does alone after the kernel."""
    try:
        fields = dict(line.split()[:2] for line in (ROOT_CGROUP / "cpu.stat").read_text().splitlines() if line.strip())
        return {"nr_throttled": int(fields["nr_throttled"]), "throttled_usec": int(fields["throttled_usec"])}
    except (OSError, ValueError, KeyError):
        return None


def oracle_frame_digest(scene: str, width: int, height: int, spp: int, max_bounces: int, seed: int, tilt: bool):
    """sha256 of the ORACLE's packed frame for this workload, from tests/golden/frame_digests.json (tools/gen_frame_digests.py:
    made in the build container, minutes of CPU; the oracle does not run here) — None if the workload has no entry."""
    try:
        digests = json.loads((ROOT / "tests" / "golden" / "frame_digests.json").read_text())
    except (OSError, ValueError):
        return None
    want = f"{scene} {width}x{height} {spp} spp max_bounces {max_bounces} seed {seed}" + (" tilt" if tilt else "")
    for entry in digests.values():
        if entry.get("workload") == want and not entry.get("partition"):
            return entry
    return None


def frame_verdict(back_buffer, entry) -> dict:
    """`frame_matches_oracle`: the frame the timed loop left in the caller's buffer against the oracle's digest."""
    import hashlib

    if entry is None:
        return {"frame_matches_oracle": None, "frame_digest_source": "no entry for this workload in tests/golden/frame_digests.json"}
    got = hashlib.sha256(back_buffer.tobytes()).hexdigest()
    return {"frame_matches_oracle": got == entry["sha256"], "frame_sha256": got[:16],
            "frame_digest_source": f"tests/golden/frame_digests.json (the oracle's frame, arithmetic contract {entry.get('contract')}, made by tools/gen_frame_digests.py); the buffer hashed is the one the timed steps rendered into"}


def mean_phases(samples: list[dict]) -> dict:
    """Average of rt_hip_phases over the timed steps (root's stream events and host clocks)."""
    if not samples:
        return {}
    out = {k: round(sum(s[k] for s in samples) / len(samples), 4) for k in PHASE_KEYS}
    if "carrier_bands" in samples[-1]:
        # the default frame mode's carrier: bands of the frame, how many of them the helper threads had carried over before the
        # stream drained (mean and the WORST step: a step whose helpers never ran shows as 0), helper threads
        out["bands"] = samples[-1]["carrier_bands"]
        out["bands_early"] = round(sum(s["carrier_bands_early"] for s in samples) / len(samples), 1)
        out["bands_early_min"] = min(s["carrier_bands_early"] for s in samples)
        out["helpers"] = samples[-1]["carrier_helpers"]
    out["transport"] = samples[-1]["transport"]
    out["scene_resident"] = int(all(s["scene_resident"] for s in samples))
    return out


def spread(values: list[float]) -> dict:
    return {"kernel_ms": [round(v, 4) for v in values], "kernel_ms_min": round(min(values), 4), "kernel_ms_max": round(max(values), 4)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="basic")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--max-bounces", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tiled", action="store_true", help="force the LDS-tiled kernel")
    ap.add_argument("--streamed", action="store_true", help="force the scalar-streamed kernel")
    ap.add_argument("--resident", action="store_true", help="force the LDS-resident kernel (what a scene of 9 to about 1300 primitives gets; scenes of up to 8 take the scalar-register kernel otherwise)")
    ap.add_argument("--tilt", action="store_true", help="single-process only: look down by 8.5 degrees from (0.2, 1.2, 3.0) instead of the scene's axis-aligned camera — the inverse view-projection then carries rounding noise in w's x / y terms (every frame of an interactive session): the general-camera build of the kernels")
    ap.add_argument("--fast", action="store_true", help="RT_HIP_FLAG_FAST: the tolerance-bound arithmetic (raw v_rsq/v_rcp), a second bench line; never the parity contract")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--frames-in-flight", type=int, default=1, help="torchrun mode with --gather torch only. 1 (default): one frame at a time, as a blocking render() caller sees it; 2: consecutive frames alternate between two streams (a throughput experiment: reported under `config`, never the default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N>1 flow on a box with fewer GPUs than ranks (frames staged through host memory)")
    ap.add_argument("--settle-ms", type=float, default=100.0, help="untimed GPU work before the W warm-up steps, so that the clocks have left idle when the timed region starts (an MI355X needs ~35 ms of load: profiles/r02/clock_ramp.txt); 0 disables")
    ap.add_argument("--no-kernel-only", action="store_true", help="skip the `kernel_only` and `plug_in_call` side legs (profiling runs: every render launch of the process is then a step of the drop-in loop)")
    ap.add_argument("--gather", default="library", choices=["library", "torch"], help="torchrun mode: `library` (default) = all three forms (see the docstring), `value` from the RCCL-gather form; `torch` = the torch form only")
    ap.add_argument("--library-deadline-s", type=float, default=0.0, help="torchrun mode: seconds the `library` form's child process may take before it is killed and reported as hung (0 = 120); the side forms get 60 s each and the three together at most --forms-budget-s")
    ap.add_argument("--forms-budget-s", type=float, default=300.0, help="torchrun mode: seconds all forms together may take; a side form that no longer fits is skipped (`status: skipped: budget`)")
    ap.add_argument("--form", default=None, choices=["library", "shared_frame", "torch"], help="(internal) this process is the child that runs one form; started by the torchrun-launched process")
    ap.add_argument("--no-interactive", action="store_true", help="skip the `interactive` side leg (the reference's operating point: ground plane in, camera tilted, frames 16 ms apart)")
    ap.add_argument("--frame-gap-ms", type=float, default=16.0, help="the `interactive` leg: host-side pause between two blocking calls (a 60 Hz caller that redraws on every frame; the GPU's clocks fall back in such gaps)")
    ap.add_argument("--direct-frame", action="store_true", help="single-process N > 1 only: RT_HIP_MULTI_DIRECT_FRAME — no gather, every GPU stores its pixels straight into the page-locked back buffer")
    ap.add_argument("--locked-frame", action="store_true", help="single-process only: time the opt-in zero-copy mode (RT_HIP_FLAG_PERSISTENT_FRAME: the caller's buffer itself page-locked and mapped) as `value` instead of the default delivery through the module's own frame")
    ap.add_argument("--same-device", action="store_true", help="single-process N > 1 only: put all N members on device 0 and move the stripes with peer copies (rehearsal on a one-GPU box; RCCL refuses duplicate devices)")
    return ap.parse_args()


# What one wave64 vector instruction costs a SIMD when it sits among plain FMAs, eight waves per SIMD — measured with
# tools/valu_rate_probe2 (profiles/r04/valu_issue_costs.txt), in shader cycles.  `int32` and `other` are blends over what the
# render loops hold of each (integer multiply 3.5, SDWA 5.1, add / shift / xor 2.2-2.6; compare and select 2.6-2.8, move 2.2).
VALU_CLASS_CYCLES = {"fma_f32": 2.22, "mul_f32": 2.22, "add_f32": 2.22, "trans_f32": 10.6, "cvt": 2.2, "int32": 2.8, "int64": 5.1, "other": 2.6}


def issue_by_class(c: dict, duration_ms: float, simds: int) -> dict | None:
    """roofline.issue.by_class: the launch's vector instructions by class (SQ_INSTS_VALU_* of a PMC pass) priced at what each
    class was MEASURED to cost, against the kernel's time: the ceiling of this instruction mix, not of a stream of FMAs."""
    names = {"fma_f32": "SQ_INSTS_VALU_FMA_F32", "mul_f32": "SQ_INSTS_VALU_MUL_F32", "add_f32": "SQ_INSTS_VALU_ADD_F32", "trans_f32": "SQ_INSTS_VALU_TRANS_F32",
             "cvt": "SQ_INSTS_VALU_CVT", "int32": "SQ_INSTS_VALU_INT32", "int64": "SQ_INSTS_VALU_INT64"}
    if any(c.get(v) is None for v in names.values()) or not c.get("SQ_INSTS_VALU"):
        return None
    counts = {k: c[v] for k, v in names.items()}
    counts["other"] = max(c["SQ_INSTS_VALU"] - sum(counts.values()), 0.0)  # compares, selects, moves, lane reads: no counter of their own
    cycles = sum(counts[k] * VALU_CLASS_CYCLES[k] for k in counts)
    ms = cycles / (simds * SHADER_CLOCK_HZ) * 1e3
    return {
        "share_of_valu_insts": {k: round(v / c["SQ_INSTS_VALU"], 4) for k, v in counts.items()},
        "cycles_per_inst_measured": VALU_CLASS_CYCLES,
        "issue_time_ms": round(ms, 4),
        "frac_of_kernel_time": round(ms / duration_ms, 4),
        "what": "every class's wave-instructions x the cycles one such instruction was measured to cost a SIMD among plain FMAs at 8 waves per SIMD (tools/valu_rate_probe2, profiles/r04/valu_issue_costs.txt), "
                "/ (1024 SIMDs x 2.4 GHz): the time this launch's vector stream needs to ISSUE, whatever else the kernel does",
    }


def make_build_line(args, pod, n_gpus, single_process):
    """The contract line for one measured form (closure over the workload)."""
    import rt_amd

    samples_total = args.width * args.height * args.spp

    def build_line(form, elapsed_s, kernels_ms, member, transport_text, more):
        ms_per_step = elapsed_s / args.steps * 1e3
        value = samples_total * args.steps / elapsed_s / 1e6
        kernel_ms = kernels_ms[0]
        # roofline of the dominant kernel: rank/member 0's launch (its share of the frame), algorithmic flops over its duration
        flops = algorithmic_flops(member["primary_samples"], member["segments"], pod.n_spheres, pod.n_planes)
        duration_ms = kernel_ms if kernel_ms == kernel_ms else ms_per_step  # NaN (two frames in flight): wall per frame
        achieved_tflops = flops / (duration_ms * 1e-3) / 1e12
        local_rows = local_rows_of(args.height, 0, n_gpus)
        scene_bytes = 20 * pod.n_spheres + 20 * pod.n_planes + 28 * pod.n_materials
        hbm_bytes = 4 * args.width * local_rows + scene_bytes
        # Counter figures of this very workload (profiles/pmc_counters.json: rocprofv3 --pmc passes of tools/gpu_profile_run.sh over
        # this command), used only if they were measured on THESE kernels: the entry carries a hash of the kernel sources, and
        # a figure from other kernels is dropped, not reported.
        traffic = None
        issue = None
        forced = "_tiled" if args.tiled else "_streamed" if args.streamed else "_resident" if args.resident else ""
        counters_key = f"{args.scene}_{args.width}x{args.height}x{args.spp}_n{n_gpus}{forced}" + ("_fast" if args.fast else "") + ("_tilt" if args.tilt else "")
        try:
            from tools.kernel_sources_hash import built_kernel_sources_sha16, kernel_sources_sha16

            built_from = built_kernel_sources_sha16() or kernel_sources_sha16()  # (the hash recorded when the library was linked)
            rec = json.loads((ROOT / "profiles" / "pmc_counters.json").read_text()).get(counters_key)
        except (ImportError, OSError, ValueError):
            built_from, rec = None, None
        if rec is not None and rec.get("kernel_sources_sha16") != built_from:
            issue = {"status": f"dropped: profiles/pmc_counters.json[{counters_key}] was measured on kernel sources {rec.get('kernel_sources_sha16')}, this build is {built_from}"}
            rec = None
        if rec is not None:
            traffic = rec.get("hbm_bytes_per_launch")
            c = rec.get("counters_per_launch", {})
            if c.get("SQ_INSTS_VALU") and duration_ms == duration_ms:
                simds = 4 * 256  # MI355X: 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies a SIMD for 2 cycles (157.3 TFLOP/s = 256 x 128 lanes x 2 flop x 2.4 GHz)
                valu, salu = c["SQ_INSTS_VALU"], c.get("SQ_INSTS_SALU", 0.0)
                ceiling_ms = valu * 2.0 / (simds * SHADER_CLOCK_HZ) * 1e3
                issue = {
                    "valu_wave_insts_per_launch": valu,
                    "salu_wave_insts_per_launch": salu,
                    "salu_per_valu": round(salu / valu, 4),
                    "waves_per_launch": c.get("SQ_WAVES"),
                    "lanes_active_frac": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4) if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None,
                    "valu_insts_per_64_samples": round(valu * 64.0 / max(member["primary_samples"], 1), 2),
                    "algorithmic_fma_insts_per_64_samples": round(flops / 2.0 / max(member["primary_samples"], 1), 2),
                    "cycles_per_valu_inst_per_simd": round(duration_ms * 1e-3 * SHADER_CLOCK_HZ * simds / valu, 3),
                    "issue_ceiling_ms": round(ceiling_ms, 4),
                    "frac_of_issue_ceiling": round(ceiling_ms / duration_ms, 4),
                    "what": "VALU wave-instructions of one launch (SQ_INSTS_VALU) x 2 cycles / (1024 SIMDs x 2.4 GHz) over this run's kernel time: how close the kernel's OWN instruction stream runs to the vector issue rate — the flop fraction above also counts against it every instruction that is not an FMA (integer hashing of the random streams, lane-mode selects, quarter-rate rsq / rcp / sqrt)",
                    "source": f"{rec.get('source')} ({rec.get('measured')}, kernel sources {rec.get('kernel_sources_sha16')}); kernel time from this run",
                }
                issue["by_class"] = issue_by_class(c, duration_ms, simds)
        roofline = {
            "bound": "valu_fp32",
            "kernel": f"render_{member['kernel']}",
            "achieved": round(achieved_tflops, 3),
            "peak": FP32_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tflops / FP32_VALU_PEAK_TFLOPS, 4),
            "traffic": traffic,
            "issue": issue,
            "kernel_ms": round(duration_ms, 4),
            "kernel_ms_source": "HIP events on the launch stream around every launch of the timed region (rt_hip_stats.render_ms), averaged" if kernel_ms == kernel_ms else "wall per frame (two frames in flight)",
            "algorithmic_flops_per_launch": flops,
            "mean_segments_per_sample": round(member["segments"] / max(member["primary_samples"], 1), 4),
            "hbm": {
                "algorithmic_bytes_per_launch": hbm_bytes,
                "achieved_GBps": round(hbm_bytes / (duration_ms * 1e-3) / 1e9, 3),
                "peak_GBps": HBM_PEAK_GBPS,
                "frac": round(hbm_bytes / (duration_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
            },
        }
        synthetic = args.scene.startswith("synthetic")
        line = {
            "metric": f"Mrays/s (W*H*spp per second) and wall-clock, {args.width}x{args.height}x{args.spp}spp scenes/{args.scene}.toml",
            "value": round(value, 1),
            "unit": "Mrays/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (generated sphere field, SURVEY.md §8d)" if synthetic else f"the reference's scene file scenes/{args.scene}.toml (no dataset; nothing is learned or loaded beyond the scene)",
            "config": {
                "workload": f"scenes/{args.scene}.toml {args.width}x{args.height} {args.spp} spp max_bounces {args.max_bounces} seed {args.seed}" + (" camera tilted (w varies over the frame)" if args.tilt else ""),
                "step": "one blocking render(): host scene in (columns fingerprinted, resident in HBM since warm-up), finished frame in the caller's host buffer out, one frame at a time",
                "spheres": pod.n_spheres,
                "planes": pod.n_planes,
                "kernel": member["kernel"],
                "arithmetic": "contract v3-fast (RT_HIP_FLAG_FAST: tolerance-bound, NOT the parity contract)" if args.fast else "contract v3 (bit-exact against the oracle)",
                "parallelism": transport_text,
                "frames_in_flight": 1,
                "frame_mode": ("locked: the caller's back buffer page-locked and mapped, kernels store straight into it (opt-in)" if args.locked_frame else "default: kernels store into the module's own page-locked frame, host threads carry the pixels into the caller's pageable back buffer while the frame is traced") if single_process else "see parallelism",
                "clock_settle_ms": args.settle_ms,
            },
            "roofline": roofline,
        }
        if form == "shared_frame":
            line["config"]["frame_buffer"] = "rank 0's back buffer is a shared mapping (MAP_SHARED) that every rank's process maps and page-locks: what the multi-process form of a direct frame needs (INTEGRATION.md)"
        line.update(more)
        return line

    return build_line


def interactive_leg(args, tracer, frame_flags) -> dict:
    """Side key `interactive`: how rt is actually used.  It renders when the back buffer is dirty — the user moves the camera
    (src/main.cpp:265-311, src/window.cpp:213-217) — so frames come a display interval apart, through a camera that is not
    axis-aligned (its inverse view-projection carries rounding noise in w: the general-camera kernels), over the scene files as
    a user has them: the ground plane of scenes/basic.toml:11-13 is one comment away.  Same size and sample count as the headline,
    the blocking drop-in call in the default frame mode, `--frame-gap-ms` of host sleep between two calls (the GPU's clocks
    fall back in the gaps).  Every figure is per call; the sleeps are not counted."""
    import numpy as np

    import rt_amd

    scene = rt_amd.Scene.named("basic_plane").set_sampling(args.spp, args.max_bounces).set_camera(*TILT)
    pod = scene.describe(args.width, args.height)
    frame = np.zeros((args.height, args.width), dtype=np.uint32)
    steps = max(1, min(args.steps, 30))
    gap = args.frame_gap_ms * 1e-3
    for _ in range(2):  # the scene change (fingerprint mismatch: one upload), the kernel's first launch
        tracer.render(pod, args.width, args.height, seed=args.seed, flags=frame_flags, out=frame)
        time.sleep(gap)
    wall_ms, kernel_ms = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        stats = tracer.render(pod, args.width, args.height, seed=args.seed, flags=frame_flags, out=frame)[2]
        wall_ms.append((time.perf_counter() - t0) * 1e3)
        kernel_ms.append(stats["render_ms"])
        time.sleep(gap)
    # the same frames back to back (no gaps): what the clocks cost
    busy = []
    for _ in range(steps):
        t0 = time.perf_counter()
        tracer.render(pod, args.width, args.height, seed=args.seed, flags=frame_flags, out=frame)
        busy.append((time.perf_counter() - t0) * 1e3)
    samples = args.width * args.height * args.spp
    flops = algorithmic_flops(stats["primary_samples"], stats["segments"], pod.n_spheres, pod.n_planes)
    mean_wall, mean_kernel = sum(wall_ms) / steps, sum(kernel_ms) / steps
    out = {
        "workload": f"scenes/basic_plane.toml (basic.toml with its commented ground plane) {args.width}x{args.height} {args.spp} spp, camera tilted (eye {TILT[0]}, looking along {TILT[1]}), one blocking call every {args.frame_gap_ms:g} ms + its own duration",
        "steps": steps,
        "ms_per_step": round(mean_wall, 4),
        "ms_per_step_min": round(min(wall_ms), 4),
        "ms_per_step_max": round(max(wall_ms), 4),
        "value": round(samples / (mean_wall * 1e-3) / 1e6, 1),
        "unit": "Mrays/s",
        "kernel_ms": round(mean_kernel, 4),
        "kernel": stats["kernel"],
        "back_to_back_ms_per_step": round(sum(busy) / steps, 4),
        "roofline_frac": round(flops / (mean_kernel * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4),
        "mean_segments_per_sample": round(stats["segments"] / max(stats["primary_samples"], 1), 4),
        "what": "the drop-in call as the plug-in's user sees it between two redraws; not `value` (BASELINE.json's metric is the plane-less, axis-aligned frame)",
    }
    out.update(frame_verdict(frame, oracle_frame_digest("basic_plane", args.width, args.height, args.spp, args.max_bounces, args.seed, True)))
    return out


def main() -> None:
    args = parse_args()
    single_process = "RANK" not in os.environ  # not launched by torchrun: one process drives all --gpus devices itself
    if single_process:
        return single_process_main(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.form:
        return form_main(args)
    return forms_parent_main(args)


def single_process_main(args) -> None:
    import numpy as np
    import torch

    import rt_amd
    from rt_amd import capi

    single_process = True
    n_gpus = args.gpus
    device = 0
    torch.cuda.set_device(device)
    scene = rt_amd.Scene.named(args.scene).set_sampling(args.spp, args.max_bounces)
    if args.tilt:
        scene.set_camera(*TILT)
    pod = scene.describe(args.width, args.height)
    flags = capi.RT_HIP_FLAG_FORCE_TILED if args.tiled else (capi.RT_HIP_FLAG_FORCE_STREAMED if args.streamed else (capi.RT_HIP_FLAG_FORCE_RESIDENT if args.resident else 0))
    if args.fast:
        flags |= capi.RT_HIP_FLAG_FAST
    samples_total = args.width * args.height * args.spp
    build_line = make_build_line(args, pod, n_gpus, True)

    def fence():
        torch.cuda.synchronize()

    def settle(render_once):
        until = time.perf_counter() + args.settle_ms * 1e-3
        while time.perf_counter() < until:  # untimed: brings the clocks up from idle; the W warm-up steps follow
            render_once()

    extras: dict = {}
    kernel_only = None
    plug_in_call = None
    if True:
        # ---- ONE process: rt_hip_render, the drop-in call, on one context (1 GPU) or one multi-GPU context ----
        if n_gpus == 1:
            tracer = rt_amd.HipRayTracer(device=device)  # fails loudly without librt_hip.so or a gfx950 device
        elif args.same_device:
            tracer = rt_amd.HipRayTracer(devices=[device] * n_gpus, peer_copy=True, direct_frame=args.direct_frame)
        else:
            tracer = rt_amd.HipRayTracer(devices=list(range(n_gpus)), direct_frame=args.direct_frame)
        back_buffer = np.zeros((args.height, args.width), dtype=np.uint32)  # rt keeps one back buffer per window size
        # the call as shim/hip_ray_tracer.cpp makes it: NO frame flag — the kernels store into the module's own page-locked
        # frame and its host threads carry the pixels into `back_buffer` (plain pageable memory) while the frame is traced.
        # --locked-frame times the opt-in zero-copy mode instead (RT_HIP_FLAG_PERSISTENT_FRAME; side key `locked_frame` otherwise)
        render_flags = flags | (capi.RT_HIP_FLAG_PERSISTENT_FRAME if args.locked_frame else 0)

        def step():
            return tracer.render(pod, args.width, args.height, seed=args.seed, flags=render_flags, out=back_buffer)[2]

        t_first = time.perf_counter()
        step()  # the very first call: scene upload, page-locking and placing the back buffer, cold clocks (reported, not timed)
        first_call_ms = (time.perf_counter() - t_first) * 1e3
        settle(step)
        for _ in range(args.warmup):
            step()
        fence()
        member_kernel_ms = [0.0] * n_gpus
        phase_samples = []
        readback_ms_sum = 0.0
        throttle0 = cgroup_throttle()
        cpu0 = process_cpu_seconds()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            stats = step()  # blocking: returns with the frame in back_buffer
            # HIP events on each member's launch stream around this step's kernel (its share of the frame)
            if n_gpus == 1:
                member_kernel_ms[0] += stats["render_ms"]
            else:
                for r in range(n_gpus):
                    member_kernel_ms[r] += tracer.member_stats(r)["render_ms"]
            readback_ms_sum += stats["readback_ms"]
            phase_samples.append(tracer.phases())
        fence()
        elapsed = time.perf_counter() - t0
        host_cpu_ms_per_step = (process_cpu_seconds() - cpu0) / args.steps * 1e3
        throttle1 = cgroup_throttle()
        member0 = tracer.member_stats(0) if n_gpus > 1 else stats
        per_rank_kernel_ms = [v / args.steps for v in member_kernel_ms]
        phases = mean_phases(phase_samples)
        transport = phases.get("transport", "none")
        extras["drop_in_breakdown"] = dict(
            {"kernel_ms": round(per_rank_kernel_ms[0], 4), "after_kernel_ms": round(readback_ms_sum / args.steps, 4), "wall_ms": round(elapsed / args.steps * 1e3, 4), "first_call_ms": round(first_call_ms, 3),
             # CPU time the whole process spent per step, ALL threads (getrusage): the caller's thread waiting in the call plus, in
             # the default frame mode, the module's helper threads polling the frame while it is traced
             "host_cpu_ms_per_step": round(host_cpu_ms_per_step, 4),
             # periods in which the job's cgroup had run out of CPU quota DURING the timed steps (every thread of the job stops
             # then, the carrier's helpers included), and for how long: null where the cgroup keeps no such record
             "cgroup_throttled": {"periods": throttle1["nr_throttled"] - throttle0["nr_throttled"], "ms": round((throttle1["throttled_usec"] - throttle0["throttled_usec"]) * 1e-3, 3)} if throttle0 and throttle1 else None},
            **{k: phases[k] for k in (*PHASE_KEYS, "bands", "bands_early", "bands_early_min", "helpers") if k in phases},
        )
        if n_gpus == 1 and not (args.tiled or args.streamed or args.resident or args.fast):
            extras.update(frame_verdict(back_buffer, oracle_frame_digest(args.scene, args.width, args.height, args.spp, args.max_bounces, args.seed, args.tilt)))
        if n_gpus > 1:
            infos = [tracer.comm_info(r) for r in range(n_gpus)]
            extras["rccl"] = {"ranks": infos[0]["ranks"], "devices": [i["device"] for i in infos], "rank_of_member": [i["rank"] for i in infos], "transport": transport, "source": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice per member" if transport == "rccl_gather" else "no communicator (this transport does not use RCCL)"}
            extras["per_rank"] = spread(per_rank_kernel_ms)

        if n_gpus == 1 and not args.no_kernel_only:
            # side figure: the call as the plug-in makes it (stats == NULL: nothing but the launch and the wait is enqueued)
            for _ in range(2):
                tracer.render(pod, args.width, args.height, seed=args.seed, flags=render_flags, out=back_buffer, stats=False)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tracer.render(pod, args.width, args.height, seed=args.seed, flags=render_flags, out=back_buffer, stats=False)
            per_frame = (time.perf_counter() - t1) / args.steps
            plug_in_call = {"ms_per_step": round(per_frame * 1e3, 4), "value": round(samples_total / per_frame / 1e6, 1), "unit": "Mrays/s", "what": "rt_hip_render with stats == NULL, as shim/hip_ray_tracer.cpp calls it: no timing events, no counter traffic (the timed steps above keep them, for the roofline's kernel time)"}
            # side figure: the other frame mode (default: the opt-in zero-copy mode; with --locked-frame: the default delivery)
            other_flags = flags | (0 if args.locked_frame else capi.RT_HIP_FLAG_PERSISTENT_FRAME)
            other_buffer = np.zeros((args.height, args.width), dtype=np.uint32)
            for _ in range(3):
                tracer.render(pod, args.width, args.height, seed=args.seed, flags=other_flags, out=other_buffer, stats=False)
            cpu1 = process_cpu_seconds()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tracer.render(pod, args.width, args.height, seed=args.seed, flags=other_flags, out=other_buffer, stats=False)
            per_frame = (time.perf_counter() - t1) / args.steps
            other_cpu_ms = (process_cpu_seconds() - cpu1) / args.steps * 1e3
            same_frame = bool(np.array_equal(other_buffer, back_buffer))
            tracer.forget_frame()
            extras["other_frame_mode"] = {"mode": "default (module-owned frame + host carrier threads)" if args.locked_frame else "locked (RT_HIP_FLAG_PERSISTENT_FRAME: the caller's buffer page-locked and mapped, zero copy; opt-in)",
                                          "ms_per_step": round(per_frame * 1e3, 4), "value": round(samples_total / per_frame / 1e6, 1), "unit": "Mrays/s", "vs_plug_in_call_ms": round(per_frame * 1e3 - plug_in_call["ms_per_step"], 4), "same_frame": same_frame, "host_cpu_ms_per_step": round(other_cpu_ms, 4),
                                          "what": "rt_hip_render with stats == NULL in the other frame mode, same scene and seed"}
            # side figure: kernel-only rate (scene resident, frame left in HBM, launches back to back)
            frame = torch.empty((args.height, args.width), dtype=torch.int32, device=f"cuda:{device}")
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                tracer.render_device(args.width, args.height, frame.data_ptr(), seed=args.seed, flags=flags, stream=stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tracer.render_device(args.width, args.height, frame.data_ptr(), seed=args.seed, flags=flags, stream=stream)
            torch.cuda.synchronize()
            per_frame = (time.perf_counter() - t1) / args.steps
            kernel_only = {"ms_per_step": round(per_frame * 1e3, 4), "value": round(samples_total / per_frame / 1e6, 1), "unit": "Mrays/s", "what": "scene resident, frame left in HBM, launches back to back (no host frame)"}
        if n_gpus == 1 and not args.no_kernel_only and not args.no_interactive and not (args.tiled or args.streamed or args.resident or args.fast):
            extras["interactive"] = interactive_leg(args, tracer, capi.RT_HIP_FLAG_PERSISTENT_FRAME if args.locked_frame else 0)
        tracers = [tracer]
        parallelism = "1 GPU" if n_gpus == 1 else f"ONE process, {n_gpus} GPUs behind rt_hip_render: row stripes of 8, transport {transport}"

    line = build_line("single", elapsed, per_rank_kernel_ms, member0, parallelism, extras)
    if plug_in_call:
        line["plug_in_call"] = plug_in_call
    if kernel_only:
        line["kernel_only"] = kernel_only
    if n_gpus == 1 and args.cpu_baseline_seconds > 0:
        line["cpu_baseline"] = cpu_baseline(args.scene, args.width, args.height, args.cpu_baseline_seconds)
    print(json.dumps(line), flush=True)
    for t in tracers:
        t.close()


# ---- one process per GPU (torchrun) ----------------------------------------------------------------------------------------------

FORMS = ("library", "shared_frame", "torch")
FORM_TEXT = {
    "library": "inside librt_hip.so: rt_hip_create + rt_hip_join_ranks (ncclCommInitRank) + one ncclGather over xGMI to rank 0, assembled into rank 0's frame; torch.distributed (gloo) only hands out the id, votes and keeps time",
    "shared_frame": "inside librt_hip.so: rt_hip_create + rt_hip_join_frame_group; every rank's kernel stores its stripes straight into ONE shared, page-locked back buffer over its own PCIe link; no data-path collective (two shared-memory counters per frame); torch.distributed (gloo) only hands out the name, votes and keeps time",
}


def local_rows_of(height: int, rank: int, world: int, stripe_rows: int = 8) -> int:
    """rt_hip_local_rows in plain Python (the torchrun-launched process does not load librt_hip.so)."""
    stripes = (height + stripe_rows - 1) // stripe_rows
    return sum(min(stripe_rows, height - b * stripe_rows) for b in range(rank, stripes, world))


def form_main(args) -> None:
    """The child process of ONE form on one rank: the only process of this rank that touches the GPU while it lives."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import rt_amd
    from rt_amd import capi, distributed

    form = args.form
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    result_path = os.environ["RT_BENCH_FORM_RESULT"]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import datetime

    if os.environ.get("RT_BENCH_DRY_RUN"):  # tests/test_bench_contract.py on a box without a GPU: the processes, ports and verdicts only
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
        everybody = distributed.all_agree(True)
        if rank == 0:
            with open(result_path, "w") as f:
                json.dump({"form": form, "status": f"dry run: {world} rank(s) met on port {os.environ['MASTER_PORT']}" if everybody else "dry run: no agreement"}, f)
        dist.barrier()
        dist.destroy_process_group()
        return
    device = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)

    # the control plane (votes, ids, barriers, clocks) is gloo; only the torch form's own data path wants RCCL from torch
    data_backend = args.backend if form == "torch" else "gloo"
    if data_backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", device), timeout=datetime.timedelta(seconds=120))  # "nccl" is RCCL on ROCm
    else:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
    vote_device = "cuda" if data_backend == "nccl" else "cpu"
    log = lambda message: print(f"[{form}] {message}", file=sys.stderr, flush=True)  # noqa: E731

    scene = rt_amd.Scene.named(args.scene).set_sampling(args.spp, args.max_bounces)
    pod = scene.describe(args.width, args.height)
    flags = capi.RT_HIP_FLAG_FORCE_TILED if args.tiled else (capi.RT_HIP_FLAG_FORCE_STREAMED if args.streamed else (capi.RT_HIP_FLAG_FORCE_RESIDENT if args.resident else 0))
    if args.fast:
        flags |= capi.RT_HIP_FLAG_FAST
    samples_total = args.width * args.height * args.spp
    result: dict = {"form": form, "status": "failed: the form's process ended without a verdict"}

    def finish(status: str, **more) -> None:
        result.clear()
        result.update({"form": form, "status": status}, **more)
        if rank == 0:
            with open(result_path + ".tmp", "w") as f:
                json.dump(result, f)
            os.replace(result_path + ".tmp", result_path)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, kernel_ms_of):
        """W warm-up steps, then exactly K steps between two fences; max over ranks; every rank's mean kernel time."""
        for _ in range(args.warmup):
            step_fn()
        fence()
        kernel_sum = 0.0
        t_begin = time.perf_counter()
        for _ in range(args.steps):
            used = step_fn()
            kernel_sum += kernel_ms_of(used)
        fence()
        local = time.perf_counter() - t_begin
        t = torch.tensor([local], dtype=torch.float64, device=vote_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernels = [None] * world
        dist.all_gather_object(kernels, kernel_sum / args.steps)
        return float(t.item()), [float(k) for k in kernels]

    def settle_collectively(step_fn):
        """Untimed steps that bring the clocks up from idle: the SAME number on every rank (the step is collective)."""
        t0 = time.perf_counter()
        step_fn()
        counts = [max(4, int(args.settle_ms * 1e-3 / max(time.perf_counter() - t0, 1e-5)))] if rank == 0 else [None]
        dist.broadcast_object_list(counts, src=0)
        for _ in range(min(int(counts[0]), 400)):
            step_fn()

    # what every form's first frame is held against: the frame rank 0 renders ALONE on its GPU (random streams are keyed by
    # the global pixel index: the frame does not depend on how it is split)
    reference = None
    if rank == 0:
        with rt_amd.HipRayTracer(device=device) as alone:
            reference = alone.render(pod, args.width, args.height, seed=args.seed, flags=flags, stats=False)[0].copy()

    tracers = []
    shared_path = None
    try:
        if form == "torch":
            in_flight = max(1, args.frames_in_flight) if args.gather == "torch" else 1
            tracers = [rt_amd.HipRayTracer(device=device) for _ in range(in_flight)]
            for t in tracers:
                t.upload(pod)
            frame_maker = distributed.DistributedFrame(tracers, args.width, args.height)
            host_frame = torch.empty((args.height, args.width), dtype=torch.int32).pin_memory() if rank == 0 else None

            def torch_step():
                frame = frame_maker.render(seed=args.seed, flags=flags)
                if frame is not None:
                    host_frame.copy_(frame, non_blocking=True)  # the frame lands in host memory, as render() must deliver it
                if in_flight == 1:
                    torch.cuda.synchronize()  # one frame at a time: what a blocking caller gets
                return tracers[(frame_maker.frames - 1) % in_flight]

            torch_step()
            torch.cuda.synchronize()
            same = True if rank != 0 else bool(np.array_equal(host_frame.numpy().view(np.uint32), reference))
            if not distributed.all_agree(same, vote_device):
                return finish("frame differs from the frame rank 0 renders alone: not used")
            settle_collectively(torch_step)
            elapsed, kernels = timed(torch_step, (lambda used: used.stats()["render_ms"]) if in_flight == 1 else (lambda used: float("nan")))
            torch.cuda.synchronize()
            transport = "torch.distributed.gather (backend nccl = RCCL)" if args.backend == "nccl" else "torch.distributed.gather over gloo (rehearsal: stripes staged through host memory)"
            return finish("ok", elapsed=elapsed, kernels=kernels, member0=tracers[(args.steps - 1) % in_flight].stats(), transport_text=transport, extras={}, frames_in_flight=in_flight)

        back_buffer = None
        if form == "shared_frame":
            names = [f"rt_hip_bench_{os.getpid()}_{int(time.time() * 1e6) & 0xFFFFFFFF:08x}"] if rank == 0 else [None]
            dist.broadcast_object_list(names, src=0)
            shared_path = f"/dev/shm/{names[0]}_frame"
            created = True
            if rank == 0:  # the "caller's back buffer": a shared mapping every rank process maps
                try:
                    with open(shared_path, "wb") as f:
                        f.truncate(args.height * args.width * 4)
                    back_buffer = np.memmap(shared_path, dtype=np.uint32, mode="r+", shape=(args.height, args.width))
                    back_buffer[:] = 0  # rt clears its back buffer before every render (src/main.cpp:318): the pages exist, first touched by rank 0
                except OSError as e:
                    log(f"rank 0: the shared frame {shared_path} could not be made: {e}")
                    created = False
            if not distributed.all_agree(created, vote_device):  # (also the barrier: the file exists before anybody maps it)
                return finish(f"not available: {shared_path} could not be created")
            tracer, reason = distributed.negotiate_rank_renderer(
                create=lambda: rt_amd.HipRayTracer(device=device),
                join=lambda t, unique: t.join_frame_group(rank, world, f"/{names[0]}_group", timeout_ms=60000),
                make_id=lambda: bytes(128),  # (nothing to hand out: the group's name is all the ranks need)
                vote_device=vote_device,
                log=log,
            )
            if tracer is None:
                return finish(f"not available: {reason}")
            tracers.append(tracer)
            if back_buffer is None:
                back_buffer = np.memmap(shared_path, dtype=np.uint32, mode="r+", shape=(args.height, args.width))
            render_flags = flags | capi.RT_HIP_FLAG_PERSISTENT_FRAME  # (implied by the group: the shared buffer is page-locked by design)
            rccl_source = "no communicator: rt_hip_join_frame_group's control block in POSIX shared memory (this transport does not use RCCL)"
        else:  # library
            if args.backend != "nccl":
                return finish("not attempted: --backend gloo rehearses the torch and the shared-frame forms only")
            tracer, reason = distributed.negotiate_rank_renderer(
                create=lambda: rt_amd.HipRayTracer(device=device),
                join=lambda t, unique: t.join_ranks(rank, world, unique, timeout_ms=60000),
                make_id=rt_amd.unique_id,
                vote_device=vote_device,
                log=log,
            )
            if tracer is None:
                return finish(f"not available: {reason}")
            tracers.append(tracer)
            back_buffer = np.zeros((args.height, args.width), dtype=np.uint32) if rank == 0 else None
            render_flags = flags  # the default delivery: the module's own page-locked frame, carried into the pageable back buffer
            rccl_source = "ncclCommCount / ncclCommUserRank / ncclCommCuDevice on every rank's communicator"

        phase_samples: list[dict] = []

        def form_step():  # collective and blocking: rank 0 returns with the frame in its back buffer
            if os.environ.get("RT_BENCH_TEST_HANG") in (form, "1"):  # tests/test_bench_contract.py: the deadline's rehearsal
                time.sleep(3600)
            return tracer.render(pod, args.width, args.height, seed=args.seed, flags=render_flags, out=back_buffer)[2]

        form_step()
        same = True if rank != 0 else bool(np.array_equal(np.asarray(back_buffer), reference))
        if not distributed.all_agree(same, vote_device):
            return finish("frame differs from the frame rank 0 renders alone: not used")

        def measured_step():
            s = form_step()
            phase_samples.append(tracer.phases())
            return s

        settle_collectively(form_step)
        elapsed, kernels = timed(measured_step, lambda s: s["render_ms"] if form != "shared_frame" else tracer.member_stats(rank)["render_ms"])
        phase_samples[:] = phase_samples[-args.steps :]
        infos = [None] * world
        dist.all_gather_object(infos, tracer.comm_info())
        phases = mean_phases(phase_samples)
        extras = {
            "drop_in_breakdown": dict({"kernel_ms": round(kernels[0], 4), "wall_ms": round(elapsed / args.steps * 1e3, 4)}, **{k: phases[k] for k in PHASE_KEYS if k in phases}),
            "rccl": {"ranks": infos[0]["ranks"], "devices": [i["device"] for i in infos], "rank_of_process": [i["rank"] for i in infos], "transport": infos[0]["transport"], "source": rccl_source},
        }
        if rank == 0:  # the frame N ranks have just assembled, against the oracle's digest of the whole frame: the same bits for every N
            extras.update(frame_verdict(np.asarray(back_buffer), oracle_frame_digest(args.scene, args.width, args.height, args.spp, args.max_bounces, args.seed, False)))
        return finish("ok", elapsed=elapsed, kernels=kernels, member0=tracer.member_stats(0), transport_text=FORM_TEXT[form], extras=extras, frames_in_flight=1)  # (member 0 = rank 0's share: the launch the roofline prices)
    except rt_amd.RtHipError as e:  # a collective renderer reports a broken frame on every rank alike
        finish(f"failed: {e}")
    finally:
        for t in tracers:
            t.close()
        try:
            dist.barrier()
        except Exception:  # noqa: BLE001 - a rank that died earlier: nothing left to wait for
            pass
        if shared_path and rank == 0:
            try:
                os.unlink(shared_path)
            except OSError:
                pass
        dist.destroy_process_group()


def forms_parent_main(args) -> None:
    """What torchrun starts on every rank: never touches a GPU.  Runs the forms one after the other, each as a child process
    per rank (own rendezvous port), kills a child that does not come back, and rank 0 prints the line."""
    import socket

    import torch.distributed as dist

    import rt_amd

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")  # CPU only: ports, verdicts
    n_gpus = args.gpus
    scene = rt_amd.Scene.named(args.scene).set_sampling(args.spp, args.max_bounces)
    pod = scene.describe(args.width, args.height)  # (host library only)
    samples_total = args.width * args.height * args.spp
    build_line = make_build_line(args, pod, n_gpus, False)
    forms = ["torch"] if args.gather == "torch" else list(FORMS)
    # The form north_star names runs FIRST and its line is printed as soon as it is done; the side forms then add their
    # figures to `paths` in a second, final line.  Every form has a deadline of its own (the one `value` comes from 120 s,
    # the side forms 60 s) and all together a budget (300 s): a driver that allows the benchmark ten minutes sees a line even
    # if every form after the first hangs.
    first_deadline = args.library_deadline_s or 120.0
    side_deadline = min(60.0, first_deadline)
    started = time.monotonic()

    def make_line(chosen, final):
        outcome = results[chosen]
        in_flight = outcome.get("frames_in_flight", 1)
        more = dict(outcome["extras"], paths=dict(paths), value_from=chosen, per_rank=spread(outcome["kernels"]) if in_flight == 1 else None)
        if "rccl" not in more:
            more["rccl"] = {"ranks": world, "devices": None, "transport": outcome["transport_text"], "source": "torch.distributed's process group (the module's own communicator was not used)"}
        line = build_line(chosen, outcome["elapsed"], outcome["kernels"], outcome["member0"], f"one process per GPU, {n_gpus} GPUs: row stripes of 8, {outcome['transport_text']}", more)
        line["config"]["frames_in_flight"] = in_flight
        line["line"] = "final" if final else f"early: printed when the `{chosen}` form was done; a final line with the side forms' `paths` follows"
        return line

    paths: dict = {}
    results: dict = {}
    any_hung = False
    early_line_from = None
    for index, form in enumerate(forms):
        deadline = first_deadline if index == 0 else side_deadline
        verdict = [index == 0 or args.forms_budget_s - (time.monotonic() - started) >= deadline]
        dist.broadcast_object_list(verdict, src=0)  # (rank 0's clock decides for everybody)
        if not verdict[0]:
            if rank == 0:
                paths[form] = {"status": f"skipped: budget ({args.forms_budget_s - (time.monotonic() - started):.0f} s of {args.forms_budget_s:.0f} s left, the form may take {deadline:.0f} s)"}
            continue
        ports = [None]
        if rank == 0:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                ports[0] = s.getsockname()[1]
        dist.broadcast_object_list(ports, src=0)
        result_path = f"/tmp/rt_bench_{os.getpid()}_{form}.json"
        # the children rendezvous among themselves on a port of their own: rank 0's child hosts that store (under torchrun the
        # workers are told to use the launcher's store — which listens on the launcher's port only)
        env = dict(os.environ, MASTER_PORT=str(ports[0]), RT_BENCH_FORM_RESULT=result_path, TORCHELASTIC_USE_AGENT_STORE="False")
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--form", form], env=env)
        hung = False
        try:
            rc = child.wait(timeout=deadline)
        except subprocess.TimeoutExpired:
            child.kill()  # this very child, by pid
            child.wait()
            rc, hung = -9, True
        verdicts = [None] * world
        dist.all_gather_object(verdicts, (rc, hung))
        hung_anywhere = any(h for _, h in verdicts)
        any_hung = any_hung or hung_anywhere
        if rank == 0:
            outcome = None
            try:
                with open(result_path) as f:
                    outcome = json.load(f)
                os.unlink(result_path)
            except (OSError, ValueError):
                pass
            if hung_anywhere:
                paths[form] = {"status": f"hung: no result within {deadline:.0f} s on rank(s) {[r for r, (_, h) in enumerate(verdicts) if h]}; the form's processes were killed"}
                for leftover in [p for p in os.listdir("/dev/shm") if p.startswith("rt_hip_bench_")]:  # what a killed form may leave behind
                    try:
                        os.unlink(os.path.join("/dev/shm", leftover))
                    except OSError:
                        pass
            elif outcome is None or any(code != 0 for code, _ in verdicts):
                paths[form] = {"status": f"failed: exit codes {[code for code, _ in verdicts]}" + (f"; {outcome['status']}" if outcome else "")}
            elif outcome["status"] != "ok":
                paths[form] = {"status": outcome["status"]}
            else:
                results[form] = outcome
                paths[form] = {"status": "ok", "ms_per_step": round(outcome["elapsed"] / args.steps * 1e3, 4), "value": round(samples_total * args.steps / outcome["elapsed"] / 1e6, 1),
                               "per_rank": spread(outcome["kernels"]) if outcome.get("frames_in_flight", 1) == 1 else None, "transport": outcome["transport_text"] if form == "torch" else outcome["extras"]["rccl"]["transport"]}
            if index == 0 and form in results and len(forms) > 1:
                print(json.dumps(make_line(form, final=False)), flush=True)
                early_line_from = form

    exit_code = 3 if any_hung else 0
    if rank == 0:
        # `value` is the form north_star names — the single RCCL gather inside the module — whenever it came up; the others are
        # side keys (`paths`).  Only if it did not: the next form that did, and `paths.library.status` says why.
        chosen = next((f for f in forms if f in results), None)
        if chosen is None:
            print(json.dumps({"metric": f"Mrays/s (W*H*spp per second) and wall-clock, {args.width}x{args.height}x{args.spp}spp scenes/{args.scene}.toml", "value": None, "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
                              "status": "no form produced a result", "paths": paths}), flush=True)
            exit_code = exit_code or 4
        else:
            line = make_line(chosen, final=True)
            if early_line_from:
                line["early_line_from"] = early_line_from
            print(json.dumps(line), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
