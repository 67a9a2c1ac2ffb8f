"""What does the destination of the pixels cost the kernel?  A rank's share of the headline frame (1/8, 1/4, 1/2, whole)
at 256 and 64 spp, rendered into HBM and into page-locked host memory on each host NUMA node (pages put there with
move_pages and checked), with the tiles the library chooses by itself: kernel time from the module's HIP events.
The buffers are registered here (hipHostRegister / hipHostGetDevicePointer through ctypes) and handed to
rt_hip_render_device, which asks HIP what kind of memory it was given."""
import ctypes, os, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
import torch
import rt_amd
from host_pages import buffer_on, host_nodes

hip = ctypes.CDLL("libamdhip64.so")
W, H = 1920, 1080
t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream
print("host NUMA nodes", host_nodes(), "; library", os.environ.get("RT_HIP_LIBRARY", "rt_amd/lib/librt_hip.so"), flush=True)


def measure(target, part):
    for _ in range(20):
        t.render_device(W, H, target, partition=part, stream=stream)
    torch.cuda.synchronize()
    ms = []
    for _ in range(40):
        t.render_device(W, H, target, partition=part, stream=stream)
        ms.append(t.stats()["render_ms"])
    return float(np.median(ms))


for spp in (256, 64):
    t.upload(rt_amd.Scene.named("basic").set_sampling(spp).describe(W, H))
    for world in (8, 4, 2, 1):
        rows = rt_amd.padded_local_rows(H, world)
        part = (0, world, 8) if world > 1 else None
        line = [f"{spp:3d} spp, share 1/{world} ({rows:4d} rows):"]
        hbm = torch.empty((rows, W), dtype=torch.int32, device="cuda:0")
        line.append(f"HBM {measure(hbm.data_ptr(), part):.4f} ms")
        for node in host_nodes():
            buf = buffer_on(node, (rows, W))
            assert hip.hipHostRegister(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes), ctypes.c_uint(2)) == 0  # hipHostRegisterMapped
            dev = ctypes.c_void_p()
            assert hip.hipHostGetDevicePointer(ctypes.byref(dev), ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint(0)) == 0
            line.append(f"host node {node} {measure(dev.value, part):.4f} ms")
            assert hip.hipHostUnregister(ctypes.c_void_p(buf.ctypes.data)) == 0
        print("   ".join(line), flush=True)
t.close()
