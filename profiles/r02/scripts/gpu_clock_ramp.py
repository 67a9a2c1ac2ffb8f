"""Kernel time of consecutive drop-in frames starting from an idle GPU: how long the clocks take to settle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch  # noqa
import rt_amd
from rt_amd import capi
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named("basic").set_sampling(256).describe(1920, 1080)
back = np.zeros((1080, 1920), dtype=np.uint32)
time.sleep(1.0)
ms = []
t0 = time.perf_counter()
for i in range(150):
    st = t.render(pod, 1920, 1080, seed=1, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)[2]
    ms.append((time.perf_counter() - t0, st["render_ms"]))
for i in (0, 1, 2, 3, 4, 5, 7, 10, 15, 20, 30, 40, 60, 80, 100, 149):
    print(f"frame {i:3d} at {ms[i][0]*1e3:7.1f} ms: kernel {ms[i][1]:.3f} ms")
