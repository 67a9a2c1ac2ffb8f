"""Samples per work item in the big-scene kernels (experiment build, RT_HIP_ITEM_SAMPLES): kernel ms by scene size, spp and
share.   RT_HIP_LIBRARY=rt_amd/lib/librt_hip_knobs.so python tools/gpu_item_sweep.py"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
import rt_amd

t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
W, H = 1920, 1080


def run(pod, part, n):
    rows = rt_amd.padded_local_rows(H, part[1]) if part else H
    frame = torch.empty((rows, W), dtype=torch.int32, device="cuda:0")
    ms = []
    for i in range(n + 1):
        t.render_device(W, H, frame.data_ptr(), partition=part, stream=stream)
        ms.append(t.stats()["render_ms"])
    return float(np.median(ms[1:]))


CASES = [(1025, 64, None, 3), (2000, 64, None, 3), (5000, 64, None, 3), (10000, 32, None, 3), (10000, 32, (0, 8, 8), 3), (30000, 64, None, 2), (100000, 64, (0, 8, 8), 2), (100000, 64, None, 1)]
for count, spp, part, n in CASES:
    pod = (rt_amd.Scene.named("synthetic-100k") if count == 100000 else rt_amd.Scene.synthetic(count)).set_sampling(spp).describe(W, H)
    t.upload(pod)
    line = [f"{count:6d} spheres x {spp} spp, {'1/8 share' if part else 'whole frame'}:"]
    for items in (16, 8, 4, 2, 1):
        if count == 100000 and not part and items in (8, 1):
            continue
        os.environ["RT_HIP_ITEM_SAMPLES"] = str(items)
        line.append(f"{items:2d}: {run(pod, part, n):8.2f}")
    print("   ".join(line), flush=True)
t.close()
