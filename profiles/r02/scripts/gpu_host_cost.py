"""Host-side cost of the calls: a 64x8 frame at 1 spp (the GPU work is nothing), with and without stats, 1 and 8 members;
and the per-frame cost of the scene check on a 100 000-sphere scene (columns fingerprinted where they lie)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import rt_amd
from rt_amd import capi

P = capi.RT_HIP_FLAG_PERSISTENT_FRAME
pod = rt_amd.Scene.named("basic").set_sampling(1).describe(64, 8)
back = np.zeros((8, 64), dtype=np.uint32)


def per_call(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


t = rt_amd.HipRayTracer(0)
t.upload(pod)
buf = torch.empty((8, 64), dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
host = per_call(lambda: t.render_device(64, 8, buf.data_ptr(), stream=s), n=2000, warm=100)
torch.cuda.synchronize()
print(f"rt_hip_render_device host time per call (tiny frame, no sync, stats kept): {host:.1f} us")
one = rt_amd.HipRayTracer(0)
print(f"single-GPU render of a tiny frame, stats kept:        {per_call(lambda: one.render(pod, 64, 8, flags=P, out=back)):.1f} us per call")
print(f"single-GPU render of a tiny frame, stats == NULL:     {per_call(lambda: one.render(pod, 64, 8, flags=P, out=back, stats=False)):.1f} us per call   (the plug-in's call)")
for name, kwargs in [("direct-frame", dict(direct_frame=True)), ("gathered", {})]:
    m = rt_amd.HipRayTracer(devices=[0] * 8, peer_copy=True, **kwargs)
    print(f"8-member {name} render of a tiny frame, stats kept:    {per_call(lambda: m.render(pod, 64, 8, flags=P, out=back)):.1f} us per call")
    print(f"8-member {name} render of a tiny frame, stats == NULL: {per_call(lambda: m.render(pod, 64, 8, flags=P, out=back, stats=False)):.1f} us per call")
    m.close()
# the headline frame: wall minus kernel
big = rt_amd.Scene.named("basic").set_sampling(256).describe(1920, 1080)
frame = np.zeros((1080, 1920), dtype=np.uint32)
for _ in range(40):
    one.render(big, 1920, 1080, flags=P, out=frame)
walls, kernels = [], []
for _ in range(30):
    t0 = time.perf_counter(); st = one.render(big, 1920, 1080, flags=P, out=frame)[2]; walls.append((time.perf_counter() - t0) * 1e3); kernels.append(st["render_ms"])
lean = per_call(lambda: one.render(big, 1920, 1080, flags=P, out=frame, stats=False), n=30, warm=5) / 1e3
print(f"headline frame: wall {np.median(walls):.4f} ms, kernel {np.median(kernels):.4f} ms (stats kept: wall - kernel = {1e3 * (np.median(walls) - np.median(kernels)):.1f} us); stats == NULL: wall {lean:.4f} ms")
# the scene check of a big scene, every frame
field = rt_amd.Scene.named("synthetic-100k").set_sampling(1).describe(64, 8)
t0 = time.perf_counter()
for _ in range(200):
    rt_amd.scene_check(field)
print(f"rt_hip_scene_check (pointer + index check + fingerprint) of 100 000 spheres: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
one.render(field, 64, 8, flags=P, out=back)
print(f"render of a tiny frame of the 100 000-sphere scene, resident: {per_call(lambda: one.render(field, 64, 8, flags=P, out=back), n=5, warm=1) / 1e3:.2f} ms per call, of which scene check {one.stats()['upload_ms']:.3f} ms")
