"""Prototype: frame rendered in row bands into HBM, band k copied to the (pinned) host frame while band k+1 traces,
against rendering straight into the mapped host frame.  Equal bands through the public partition {b, B, band_rows}."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import rt_amd
from rt_amd import capi

W, H, SPP = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named("basic").set_sampling(SPP).describe(W, H)
t.upload(pod)
want = t.render(pod, W, H, seed=1)[0]
host = torch.empty((H, W), dtype=torch.int32).pin_memory()
frame = torch.empty((H, W), dtype=torch.int32, device="cuda:0")
S = [torch.cuda.Stream(), torch.cuda.Stream()]
C = torch.cuda.Stream()

def banded(B):
    rows = -(-H // B // 8) * 8
    order = list(range(B))[::-1]  # bottom band first
    evs = []
    for i, b in enumerate(order):
        s = S[i % 2]
        r0 = b * rows
        n = min(rows, H - r0)
        t.render_device(W, H, frame[r0:].data_ptr(), seed=1, partition=(b, B, rows), stream=s.cuda_stream)
        e = torch.cuda.Event(); e.record(s)
        C.wait_event(e)
        with torch.cuda.stream(C):
            host[r0:r0 + n].copy_(frame[r0:r0 + n], non_blocking=True)
    C.synchronize()

back = np.zeros((H, W), dtype=np.uint32)
def mapped():
    t.render(pod, W, H, seed=1, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)

def timeit(f, n=40):
    for _ in range(12): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2], sum(ts) / len(ts), ts[-1]

for rnd in range(2):
    print("mapped        min %.3f med %.3f mean %.3f max %.3f" % timeit(mapped))
    for B in (2, 3, 4, 6):
        print("banded B=%d    min %.3f med %.3f mean %.3f max %.3f" % ((B,) + timeit(lambda: banded(B))))
        assert np.array_equal(host.numpy().view(np.uint32), want), B
