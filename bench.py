#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (= W*H*spp / wall-seconds / 1e6) and wall-clock for a 1920x1080x256spp render of
scenes/basic.toml (BASELINE.json `metric`), on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full frame: every rank renders its row stripes with the HIP kernels (scene already resident in
HBM), the stripes are gathered to rank 0 over RCCL and de-interleaved there (N = 1: kernel only).  The frame is
fixed as N grows, so scaling is STRONG.  Rank 0 prints ONE JSON line.

The `roofline` object prices the render kernel against the FP32 vector-ALU peak — the bound SURVEY.md §8d
identifies for this path (a 3-sphere scene is ~100 bytes; the only compulsory HBM traffic is the 4 B/pixel frame) —
and carries the HBM figures next to it.  `cpu_baseline` times the reference-faithful CPU model (oracle/, mt19937
mode) on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP32_VALU_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (vector)"
HBM_PEAK_GBPS = 8000.0  # same guide, HBM3E spec


def algorithmic_flops(samples: int, segments: int, n_spheres: int, n_planes: int) -> float:
    """S*70 + segments*(22*N_s + 16*N_p + 60)."""
    return samples * 70.0 + segments * (22.0 * n_spheres + 16.0 * n_planes + 60.0)


def cpu_baseline(scene_name: str, width: int, height: int, target_seconds: float) -> dict:
    """Reference-faithful CPU model (oracle, mt19937 mode) on all host cores this process may use, on a bounded
    sample: the same scene and frame size at a reduced spp chosen to take about `target_seconds`."""
    import rt_amd
    from oracle import binding as oracle  # cpu_baseline leg: the oracle is the thing timed here, by design

    cores = len(os.sched_getaffinity(0))
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
        "sample": f"{scene_name}.toml {width}x{height} at {spp} spp (of the workload's spp), mt19937 model, -O3 -mavx2 -mfma -ffast-math, {stats['seconds']:.1f} s",
    }


def main() -> None:
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
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="1: one frame at a time; 2: consecutive frames alternate between two streams; 0 = 1 on one GPU, 2 on several")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N>1 flow on a box with fewer GPUs than ranks (frames staged through host memory)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import rt_amd
    from rt_amd import capi, distributed

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    device = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    in_flight = args.frames_in_flight or (1 if world == 1 else 2)
    tracers = [rt_amd.HipRayTracer(device=device) for _ in range(in_flight)]  # fails loudly without librt_hip.so or a gfx950 device
    tracer = tracers[0]
    scene = rt_amd.Scene.named(args.scene).set_sampling(args.spp, args.max_bounces)
    pod = scene.describe(args.width, args.height)
    for t in tracers:
        t.upload(pod)  # inputs resident in HBM before the timed region
    flags = capi.RT_HIP_FLAG_FORCE_TILED if args.tiled else (capi.RT_HIP_FLAG_FORCE_STREAMED if args.streamed else 0)
    frame = distributed.DistributedFrame(tracers, args.width, args.height)

    def step():
        return frame.render(seed=args.seed, flags=flags)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()

    # kernel duration: HIP events on the launch stream around every render launch of the timed region
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    launch = {"i": 0}

    def timed(real_render_device):
        def timed_render_device(*a, **k):
            i = launch["i"]
            stream = torch.cuda.current_stream()  # DistributedFrame launches on the current stream of its slot
            starts[i].record(stream)
            real_render_device(*a, **k)
            ends[i].record(stream)
            launch["i"] = i + 1

        return timed_render_device

    originals = [t.render_device for t in tracers]
    for t, original in zip(tracers, originals):
        t.render_device = timed(original)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    for t, original in zip(tracers, originals):
        t.render_device = original

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps
    stats = tracers[(args.steps - 1) % in_flight].stats()  # counters of this rank's last launch

    if rank == 0:
        samples_total = args.width * args.height * args.spp
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_total * args.steps / elapsed / 1e6

        flops = algorithmic_flops(stats["primary_samples"], stats["segments"], pod.n_spheres, pod.n_planes)
        # one frame at a time: the launch duration from the HIP events.  Two frames in flight share the GPU, so a
        # launch's own begin-to-end time says nothing about its rate: use the wall time per frame instead.
        duration_ms = kernel_ms if in_flight == 1 else ms_per_step
        achieved_tflops = flops / (duration_ms * 1e-3) / 1e12
        local_rows = rt_amd.local_rows(args.height, 0, world)
        scene_bytes = 20 * pod.n_spheres + 20 * pod.n_planes + 28 * pod.n_materials
        hbm_bytes = 4 * args.width * local_rows + scene_bytes
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                rec = json.loads(pmc.read_text())
                key = f"{args.scene}_{args.width}x{args.height}x{args.spp}_n{world}"
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except (ValueError, OSError):
                traffic = None
        roofline = {
            "bound": "valu_fp32",
            "kernel": f"render_{stats['kernel']}",
            "achieved": round(achieved_tflops, 3),
            "peak": FP32_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved_tflops / FP32_VALU_PEAK_TFLOPS, 4),
            "traffic": traffic,
            "kernel_ms": round(kernel_ms, 4),
            "duration_used_ms": round(duration_ms, 4),
            "algorithmic_flops_per_launch": flops,
            "mean_segments_per_sample": round(stats["segments"] / max(stats["primary_samples"], 1), 4),
            "hbm": {
                "algorithmic_bytes_per_launch": hbm_bytes,
                "achieved_GBps": round(hbm_bytes / (duration_ms * 1e-3) / 1e9, 3),
                "peak_GBps": HBM_PEAK_GBPS,
                "frac": round(hbm_bytes / (duration_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
            },
        }
        line = {
            "metric": f"Mrays/s (W*H*spp per second) and wall-clock, {args.width}x{args.height}x{args.spp}spp scenes/{args.scene}.toml",
            "value": round(value, 1),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"scenes/{args.scene}.toml {args.width}x{args.height} {args.spp} spp max_bounces {args.max_bounces} seed {args.seed}",
                "spheres": pod.n_spheres,
                "planes": pod.n_planes,
                "kernel": stats["kernel"],
                "parallelism": f"row stripes of 8 over {world} GPU(s)" + (" + 1 RCCL gather to rank 0 + device assemble" if world > 1 else ""),
                "frames_in_flight": in_flight,
            },
            "roofline": roofline,
        }
        if world == 1:
            # the drop-in call as rt makes it: host scene in, host frame out (upload + kernel + PCIe read-back); never `value`
            import numpy as np

            back_buffer = np.zeros((args.height, args.width), dtype=np.uint32)  # rt keeps one back buffer per window size
            walls = []
            for _ in range(5):
                t0 = time.perf_counter()
                tracer.render(pod, args.width, args.height, seed=args.seed, flags=flags | capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back_buffer)
                walls.append(time.perf_counter() - t0)
            wall = sorted(walls[1:])[len(walls[1:]) // 2]  # the first call page-locks the buffer and uploads the scene
            line["drop_in_render"] = {"wall_ms": round(wall * 1e3, 3), "value": round(samples_total / wall / 1e6, 1), "unit": "Mrays/s", "includes": "scene fingerprint check (upload skipped when unchanged) + kernel + D2H of the frame into the caller's page-locked back buffer"}
        if world == 1 and args.cpu_baseline_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.scene, args.width, args.height, args.cpu_baseline_seconds)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for t in tracers:
        t.close()


if __name__ == "__main__":
    main()
