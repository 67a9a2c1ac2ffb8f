#!/usr/bin/env python3
"""Kernel time of one rank's share of the headline frame for world sizes 1, 2, 4, 8 (all ranks, on ONE GPU, one after the
other): how much of the ideal 1/N the render kernel keeps when the frame gets small.  usage: python tools/gpu_partition_times.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import rt_amd  # noqa: E402

width, height, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = rt_amd.Scene.named("basic").set_sampling(spp)
tracer = rt_amd.HipRayTracer(0)
tracer.upload(scene.describe(width, height))
whole = None
for world in (1, 2, 4, 8):
    rows = rt_amd.padded_local_rows(height, world, 8)
    buf = torch.empty((rows, width), dtype=torch.int32, device="cuda:0")
    per_rank = []
    for rank in range(world):
        best = 1e9
        for _ in range(6):
            tracer.render_device(width, height, buf.data_ptr(), seed=1, partition=(rank, world, 8))
            best = min(best, tracer.stats()["render_ms"])
        per_rank.append(best)
    slowest = max(per_rank)
    whole = whole or slowest
    print(f"world {world}: slowest rank {slowest:.3f} ms, fastest {min(per_rank):.3f} ms, ideal {whole / world:.3f} ms -> kernel-only efficiency {whole / world / slowest:.3f}")

# the same rank-sized frames back to back on two streams (two contexts), as bench.py runs them on N > 1: wall per frame
tracers = [tracer, rt_amd.HipRayTracer(0)]
tracers[1].upload(scene.describe(width, height))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for world in (2, 4, 8):
    rows = rt_amd.padded_local_rows(height, world, 8)
    bufs = [torch.empty((rows, width), dtype=torch.int32, device="cuda:0") for _ in range(2)]
    for in_flight in (1, 2):
        frames = 60
        torch.cuda.synchronize()
        import time

        t0 = time.perf_counter()
        for k in range(frames):
            slot = k % in_flight
            tracers[slot].render_device(width, height, bufs[slot].data_ptr(), seed=1, partition=(0, world, 8), stream=streams[slot].cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / frames
        print(f"world {world}, {in_flight} in flight: {ms:.3f} ms per frame (ideal {whole / world:.3f})")
