import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import rt_amd
from rt_amd import capi
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named("basic").set_sampling(1).describe(64, 8)
t.upload(pod)
buf = torch.empty((8, 64), dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream().cuda_stream
for _ in range(100): t.render_device(64, 8, buf.data_ptr(), stream=s)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 2000
for _ in range(n): t.render_device(64, 8, buf.data_ptr(), stream=s)
host = (time.perf_counter() - t0) / n * 1e6
torch.cuda.synchronize()
print(f"rt_hip_render_device host time per call (tiny frame, no sync): {host:.1f} us")
m = rt_amd.HipRayTracer(devices=[0]*8, peer_copy=True, direct_frame=True)
back = np.zeros((8, 64), dtype=np.uint32)
for _ in range(20): m.render(pod, 64, 8, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)
t0 = time.perf_counter()
for _ in range(300): m.render(pod, 64, 8, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)
print(f"8-member direct-frame render of a tiny frame: {(time.perf_counter()-t0)/300*1e6:.1f} us per call")
g = rt_amd.HipRayTracer(devices=[0]*8, peer_copy=True)
for _ in range(20): g.render(pod, 64, 8, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)
t0 = time.perf_counter()
for _ in range(300): g.render(pod, 64, 8, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)
print(f"8-member gathered render of a tiny frame: {(time.perf_counter()-t0)/300*1e6:.1f} us per call")
one = rt_amd.HipRayTracer(0)
for _ in range(20): one.render(pod, 64, 8, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)
t0 = time.perf_counter()
for _ in range(300): one.render(pod, 64, 8, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)
print(f"single-GPU render of a tiny frame: {(time.perf_counter()-t0)/300*1e6:.1f} us per call")
