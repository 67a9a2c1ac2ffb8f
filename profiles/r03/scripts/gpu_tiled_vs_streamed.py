"""Big scenes at low sample counts: the LDS-tiled kernel (the default below 32 spp since round 1) against the streamed one
as it is now; kernel ms at 1920x1080, frame in HBM."""
import sys
sys.path.insert(0, ".")
import torch
import rt_amd
from rt_amd import capi

t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
W, H = 1920, 1080
frame = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
for count in (1100, 2000, 10000, 100000):
    for spp in (1, 2, 4, 8, 16, 24):
        if count == 100000 and spp > 8:
            continue
        scene = rt_amd.Scene.named("synthetic-100k") if count == 100000 else rt_amd.Scene.synthetic(count)
        t.upload(scene.set_sampling(spp).describe(W, H))
        line = [f"{count:6d} spheres x {spp:2d} spp:"]
        for name, flags in (("tiled", capi.RT_HIP_FLAG_FORCE_TILED), ("streamed", capi.RT_HIP_FLAG_FORCE_STREAMED)):
            ms = []
            for _ in range(3 if count < 100000 else 2):
                t.render_device(W, H, frame.data_ptr(), flags=flags, stream=stream)
                ms.append(t.stats()["render_ms"])
            line.append(f"{name} {min(ms[1:]):9.2f}")
        print("   ".join(line), flush=True)
t.close()
