"""Small frames of big scenes: kernel ms of the streamed kernel (the library's choice: sparse launches spread thin) — and, with
RT_HIP_LIBRARY pointing at a build without the lane cap, the same before."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import rt_amd

t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
for count in (100000, 10000, 2000):
    scene = rt_amd.Scene.named("synthetic-100k") if count == 100000 else rt_amd.Scene.synthetic(count)
    for w, h, spp in ((64, 8, 1), (64, 36, 1), (64, 36, 4), (160, 90, 1), (160, 90, 4), (320, 180, 1), (320, 180, 4), (640, 360, 1)):
        t.upload(scene.set_sampling(spp).describe(w, h))
        frame = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
        ms = []
        for _ in range(4):
            t.render_device(w, h, frame.data_ptr(), stream=stream)
            ms.append(t.stats()["render_ms"])
        print(f"{count:6d} spheres, {w}x{h} at {spp} spp ({w * h * spp} samples): {min(ms[1:]):8.3f} ms ({t.stats()['kernel']})", flush=True)
t.close()
