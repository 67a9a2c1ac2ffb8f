"""Timing experiment for half-chunks: what would 8-sample work items buy a short launch?
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_chunk8.so is a build whose chunks are 8 samples (its frames are NOT the contract's; an
upper bound for a half-chunk kernel, which would also have to park the second half's sample values); the knobs build has the
contract's 16.  Shares of the headline frame at 64 and 256 spp, into HBM, tile sizes from the environment."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
import rt_amd

W, H = 1920, 1080
t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
print("library", os.environ.get("RT_HIP_LIBRARY"), flush=True)


def measure(target, part):
    for _ in range(20):
        t.render_device(W, H, target, partition=part, stream=stream)
    torch.cuda.synchronize()
    ms = []
    for _ in range(40):
        t.render_device(W, H, target, partition=part, stream=stream)
        ms.append(t.stats()["render_ms"])
    return float(np.median(ms))


for spp in (64, 256):
    t.upload(rt_amd.Scene.named("basic").set_sampling(spp).describe(W, H))
    for world in (8, 4, 2, 1):
        rows = rt_amd.padded_local_rows(H, world)
        part = (0, world, 8) if world > 1 else None
        hbm = torch.empty((rows, W), dtype=torch.int32, device="cuda:0")
        line = [f"{spp:3d} spp, share 1/{world}:"]
        for p in (None, 2, 3, 4, 5):
            if p is None:
                os.environ.pop("RT_HIP_TILE_LOG2", None)
            else:
                os.environ["RT_HIP_TILE_LOG2"] = str(p)
            line.append(f"{'auto' if p is None else str(1 << p) + ' px'} {measure(hbm.data_ptr(), part):.4f}")
        print("   ".join(line), flush=True)
t.close()
