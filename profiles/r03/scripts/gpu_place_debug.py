"""Why do moved pages misbehave?  (a) shm mapping: read-touch vs write-touch, then move_pages: status codes.
(b) a direct frame of 4 members on one device into np.zeros memory: RT_HIP_NUMA_MOVE=0 against the library's placement."""
import ctypes, mmap, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
from host_pages import nodes_of, host_nodes, _pages, _libc, SYS_MOVE_PAGES, MPOL_MF_MOVE

W, H = 1920, 1080


def move_status(buf, node):
    pages, count = _pages(buf)
    nodes = (ctypes.c_int * count)(*([node] * count))
    status = (ctypes.c_int * count)()
    rc = _libc.syscall(SYS_MOVE_PAGES, 0, ctypes.c_ulong(count), pages, nodes, status, MPOL_MF_MOVE)
    vals, counts = np.unique(np.array(status[:]), return_counts=True)
    return rc, ctypes.get_errno(), {int(v): int(c) for v, c in zip(vals, counts)}


for touch in ("read", "write"):
    path = f"/dev/shm/rt_hip_place_debug_{os.getpid()}"
    with open(path, "wb") as f:
        f.truncate(W * H * 4)
    other = np.memmap(path, dtype=np.uint32, mode="r+", shape=(H, W))
    other[:] = 0
    del other
    buf = np.memmap(path, dtype=np.uint32, mode="r+", shape=(H, W))
    flat = buf.reshape(-1)
    before = nodes_of(buf)
    if touch == "read":
        s = int(flat[::1024].sum())
    else:
        flat[::1024] |= 0
    touched = nodes_of(buf)
    target = [n for n in host_nodes() if n not in touched][:1] or [host_nodes()[-1]]
    print(f"shm, {touch}-touch: before {before}, touched {touched}; move to node {target[0]}:", move_status(buf, target[0]), "->", nodes_of(buf), flush=True)
    del buf, flat
    os.unlink(path)

import rt_amd
from rt_amd import capi
P = capi.RT_HIP_FLAG_PERSISTENT_FRAME
pod = rt_amd.Scene.named("basic").set_sampling(256).describe(W, H)
for members in (1, 4):
    for knob in ("0", None):
        if knob is None:
            os.environ.pop("RT_HIP_NUMA_MOVE", None)
        else:
            os.environ["RT_HIP_NUMA_MOVE"] = knob
        t = rt_amd.HipRayTracer(0) if members == 1 else rt_amd.HipRayTracer(devices=[0] * members, peer_copy=True, direct_frame=True)
        for kind in ("np.zeros", "written"):
            frame = np.zeros((H, W), dtype=np.uint32)
            if kind == "written":
                frame.fill(1)
            first = nodes_of(frame)
            t.render(pod, W, H, flags=P, out=frame)
            for _ in range(30):
                t.render(pod, W, H, flags=P, out=frame)
            walls = []
            for _ in range(20):
                t0 = time.perf_counter(); t.render(pod, W, H, flags=P, out=frame, stats=False); walls.append((time.perf_counter() - t0) * 1e3)
            print(f"{members} member(s), RT_HIP_NUMA_MOVE={knob}, {kind}: pages {first} -> {nodes_of(frame)}; wall {np.median(walls):.4f} ms", flush=True)
            t.forget_frame()
        t.close()
