"""Soak of rank shares of small scenes: random scenes (0-12 spheres, 0-3 planes), random frames, random sample counts (1-139), a
random rank of a random world of 1-8, random kernel / item flags, frame in HBM — against the oracle's rows of the same
partition, bit for bit.      python tools/gpu_partition_soak.py [cases]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import rt_amd
from oracle import binding as oracle
from rt_amd import capi
from tests.test_gpu_parity import random_scene

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
KERNELS = (0, capi.RT_HIP_FLAG_FORCE_RESIDENT, capi.RT_HIP_FLAG_FORCE_STREAMED, capi.RT_HIP_FLAG_FORCE_TILED)
ITEMS = (0, capi.RT_HIP_FLAG_FORCE_HALF_CHUNKS, capi.RT_HIP_FLAG_FORCE_WHOLE_CHUNKS)
bad = 0
for case in range(cases):
    rng = np.random.default_rng(50000 + case)
    spheres, planes, materials, camera = random_scene(rng)
    width, height = int(rng.integers(1, 160)), int(rng.integers(1, 100))
    spp, bounces = int(rng.integers(1, 140)) if case % 3 == 0 else int(rng.integers(1, 33)), int(rng.integers(1, 12))
    world = int(rng.integers(1, 9)); rank = int(rng.integers(0, world))
    ivp = camera.describe(width, height).inverse_view_projection[:]
    pod = rt_amd.scene_from_arrays(spheres, planes, materials, samples_per_pixel=spp, max_bounces=bounces, inverse_view_projection=ivp)
    seed = int(rng.integers(0, 2**63))
    part = (rank, world, 8)
    want, want_rgb, want_stats = oracle.render(pod, width, height, seed=seed, partition=part)
    rows = rt_amd.padded_local_rows(height, world)
    t.upload(pod)
    for flags in (int(rng.choice(KERNELS)) | int(rng.choice(ITEMS)), int(rng.choice(KERNELS)) | int(rng.choice(ITEMS))):
        d_rgba = torch.zeros((max(rows, 1), width), dtype=torch.int32, device="cuda:0")
        d_rgb = torch.zeros((max(rows, 1), width, 3), dtype=torch.float32, device="cuda:0")
        t.render_device(width, height, d_rgba.data_ptr(), seed=seed, flags=flags, partition=part, d_rgb_f32=d_rgb.data_ptr(), stream=stream)
        stats = t.stats()
        got = d_rgba.cpu().numpy().view(np.uint32)[: want.shape[0]]
        rgb = d_rgb.cpu().numpy()[: want.shape[0]]
        same = (rgb.view(np.uint32) == want_rgb.view(np.uint32)) | (np.isnan(rgb) & np.isnan(want_rgb))
        if not (np.array_equal(got, want) and same.all() and stats["segments"] == want_stats["segments"]):
            bad += 1
            print(f"MISMATCH case {case}: {len(spheres)} spheres {len(planes)} planes {width}x{height}x{spp} rank {rank}/{world} kernel {stats['kernel']} flags {flags}: {(got != want).sum()} words, {(~same).sum()} floats", flush=True)
    if case % 250 == 249:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"DONE: {cases} cases x 2 random modes, {bad} mismatches")
t.close()
sys.exit(1 if bad else 0)
