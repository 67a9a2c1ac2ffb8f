"""Soak of the big-scene kernels' item forms: random sphere fields (1 025 .. 4 000 spheres), tiny random frames, random sample
counts, a random rank of a random world, streamed or tiled, the launch code's item size / eight samples forced / whole chunks
forced, frame in HBM — against the oracle's rows of the same partition, bit for bit (RGBA8 and the float mean).
    python tools/gpu_big_scene_soak.py [cases]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import rt_amd
from oracle import binding as oracle
from rt_amd import capi
from tests.test_gpu_parity import _sphere_field

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
bad = 0
for case in range(cases):
    rng = np.random.default_rng(90000 + case)
    count = int(rng.integers(1025, 4000))
    width, height = int(rng.integers(3, 48)), int(rng.integers(2, 40))
    spp, bounces = int(rng.integers(1, 72)), int(rng.integers(1, 9))
    world = int(rng.integers(1, 5)); rank = int(rng.integers(0, world))
    spheres, materials, camera = _sphere_field(rng, count)
    ivp = camera.describe(width, height).inverse_view_projection[:]
    pod = rt_amd.scene_from_arrays(spheres, [], materials, samples_per_pixel=spp, max_bounces=bounces, inverse_view_projection=ivp)
    seed = int(rng.integers(0, 2**63))
    part = (rank, world, 8)
    want, want_rgb, want_stats = oracle.render(pod, width, height, seed=seed, partition=part)
    rows = rt_amd.padded_local_rows(height, world)
    t.upload(pod)
    for kernel in (capi.RT_HIP_FLAG_FORCE_STREAMED, capi.RT_HIP_FLAG_FORCE_TILED):
        for items in (0, capi.RT_HIP_FLAG_FORCE_HALF_CHUNKS, capi.RT_HIP_FLAG_FORCE_WHOLE_CHUNKS):
            d_rgba = torch.zeros((rows, width), dtype=torch.int32, device="cuda:0")
            d_rgb = torch.zeros((rows, width, 3), dtype=torch.float32, device="cuda:0")
            t.render_device(width, height, d_rgba.data_ptr(), seed=seed, flags=kernel | items, partition=part, d_rgb_f32=d_rgb.data_ptr(), stream=stream)
            stats = t.stats()
            got = d_rgba.cpu().numpy().view(np.uint32)[: want.shape[0]]
            rgb = d_rgb.cpu().numpy()[: want.shape[0]]
            same = (rgb.view(np.uint32) == want_rgb.view(np.uint32)) | (np.isnan(rgb) & np.isnan(want_rgb))
            if not (np.array_equal(got, want) and same.all() and stats["segments"] == want_stats["segments"]):
                bad += 1
                print(f"MISMATCH case {case}: {count} spheres {width}x{height}x{spp} rank {rank}/{world} kernel {stats['kernel']} items flag {items}: {(got != want).sum()} words, {(~same).sum()} floats", flush=True)
    if case % 25 == 24:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"DONE: {cases} cases x 6 modes, {bad} mismatches")
t.close()
sys.exit(1 if bad else 0)
