"""Config 5 (100 000 spheres, 1920x1080x64) and its 1/8 share with whole chunks against sub-chunk items, and two mid-size
big scenes: kernel ms from the module's events (frame left in HBM)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import rt_amd
from rt_amd import capi

t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
W, H = 1920, 1080
WHOLE, HALF = capi.RT_HIP_FLAG_FORCE_WHOLE_CHUNKS, capi.RT_HIP_FLAG_FORCE_HALF_CHUNKS


def run(label, pod, part, flags, n):
    rows = rt_amd.padded_local_rows(H, part[1]) if part else H
    frame = torch.empty((rows, W), dtype=torch.int32, device="cuda:0")
    t.upload(pod)
    ms = []
    for i in range(n + 1):
        t.render_device(W, H, frame.data_ptr(), flags=flags, partition=part, stream=stream)
        ms.append(t.stats()["render_ms"])
    print(f"{label}: {np.median(ms[1:]):.2f} ms (runs {[round(m, 1) for m in ms]})", flush=True)


big = rt_amd.Scene.named("synthetic-100k").set_sampling(64).describe(W, H)
for name, flags in (("whole chunks", WHOLE), ("the library's choice", 0)):
    run(f"config 5, 1/8 share, {name}", big, (0, 8, 8), flags, 2)
for name, flags in (("whole chunks", WHOLE), ("the library's choice", 0)):
    run(f"config 5, whole frame, {name}", big, None, flags, 2)
for count, spp in ((10000, 32), (2000, 64)):
    pod = rt_amd.Scene.synthetic(count).set_sampling(spp).describe(W, H)
    for name, flags in (("whole chunks", WHOLE), ("the library's choice", 0)):
        run(f"{count} spheres x {spp} spp, {name}", pod, None, flags, 3)
t.close()
