"""Mid-size scenes (9 .. 2000 spheres): the LDS-resident kernel (one tile per wave) against the scalar-streamed and the
LDS-tiled rolling kernels; kernel ms at 1920x1080, frame in HBM."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import rt_amd
from rt_amd import capi

t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
W, H = 1920, 1080
frame = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
for spp in (64, 16):
    for count in (12, 32, 64, 128, 256, 512, 1024, 2000):
        pod = rt_amd.Scene.synthetic(count).set_sampling(spp).describe(W, H)
        t.upload(pod)
        line = [f"{count:5d} spheres x {spp} spp:"]
        for name, flags in (("default", 0), ("resident", capi.RT_HIP_FLAG_FORCE_RESIDENT), ("streamed", capi.RT_HIP_FLAG_FORCE_STREAMED), ("tiled", capi.RT_HIP_FLAG_FORCE_TILED)):
            if name == "resident" and count > 1024:
                continue
            ms = []
            for _ in range(3):
                t.render_device(W, H, frame.data_ptr(), flags=flags, stream=stream)
                ms.append(t.stats()["render_ms"])
            line.append(f"{name} {min(ms[1:]):8.2f} ({t.stats()['kernel']})")
        print("   ".join(line), flush=True)
t.close()
