"""Pixel-tile size and shape of the small kernel against the destination of the pixels (experiment build
librt_hip_knobs.so: RT_HIP_TILE_LOG2 / RT_HIP_TILE_W_LOG2 per launch).  A wave stores its tile's finished pixels as row
fragments of tile_w pixels; into page-locked host memory every fragment is a PCIe write of 4 x tile_w bytes.
    RT_HIP_LIBRARY=rt_amd/lib/librt_hip_knobs.so python tools/gpu_tile_shapes.py
For the headline frame's 1/8, 1/4 and whole shares at 256 and 64 spp: kernel ms into HBM and into near host memory."""
import ctypes, os, sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
import torch
import rt_amd

hip = ctypes.CDLL("libamdhip64.so")


from host_pages import buffer_on, host_nodes

NODES = host_nodes()
W, H = 1920, 1080
t = rt_amd.HipRayTracer(0)
stream = torch.cuda.current_stream().cuda_stream


def measure(target, part, n=30):
    for _ in range(15):
        t.render_device(W, H, target, partition=part, stream=stream)
    torch.cuda.synchronize()
    ms = []
    for _ in range(n):
        t.render_device(W, H, target, partition=part, stream=stream)
        ms.append(t.stats()["render_ms"])
    return float(np.median(ms))


for spp in (256, 64):
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(W, H)
    t.upload(pod)
    for world in (8, 4, 2, 1):
        rows = rt_amd.padded_local_rows(H, world)
        part = (0, world, 8) if world > 1 else None
        hosts, devs = [], []
        for node in NODES:  # one buffer per host NUMA node
            host = buffer_on(node, (rows, W))
            assert hip.hipHostRegister(ctypes.c_void_p(host.ctypes.data), ctypes.c_size_t(host.nbytes), ctypes.c_uint(2)) == 0
            dev = ctypes.c_void_p()
            assert hip.hipHostGetDevicePointer(ctypes.byref(dev), ctypes.c_void_p(host.ctypes.data), ctypes.c_uint(0)) == 0
            hosts.append(host)
            devs.append(dev.value)
        hbm = torch.empty((rows, W), dtype=torch.int32, device="cuda:0")
        print(f"--- {spp} spp, share 1/{world} ({rows} rows)", flush=True)
        shapes = [(None, None)] + [(p, w) for p in ((2, 3, 4) if spp == 256 else (4, 5)) for w in range((p + 1) // 2, min(p, 4) + 1)]
        for p, w in shapes:
            for key, val in (("RT_HIP_TILE_LOG2", p), ("RT_HIP_TILE_W_LOG2", w)):
                if val is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = str(val)
            label = "as chosen by the library" if p is None else f"{1 << p:3d} pixels as {1 << w:2d} x {(1 << p) >> w}"
            print(f"  {label:28s} HBM {measure(hbm.data_ptr(), part):.4f} ms   " + "   ".join(f"host node {node} {measure(dev, part):.4f} ms" for node, dev in zip(NODES, devs)), flush=True)
        for host in hosts:
            assert hip.hipHostUnregister(ctypes.c_void_p(host.ctypes.data)) == 0
t.close()
