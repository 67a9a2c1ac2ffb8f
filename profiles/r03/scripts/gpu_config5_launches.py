"""Consecutive drop-in frames of BASELINE config 5 (synthetic 100 000 spheres, 1920x1080, 64 spp) with a wall-clock
stamp per launch, so that tools/gpu_clock_sampler.py's clock/power trace can be laid next to the kernel times.

    python tools/gpu_config5_launches.py [frames] [pause_s] [kernel: streamed|tiled]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch  # noqa: F401  (torch first: capi.hip_lib's load order)
import rt_amd
from rt_amd import capi

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
flags = capi.RT_HIP_FLAG_PERSISTENT_FRAME | (capi.RT_HIP_FLAG_FORCE_TILED if len(sys.argv) > 3 and sys.argv[3] == "tiled" else 0)
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named("synthetic-100k").set_sampling(64).describe(1920, 1080)
back = np.zeros((1080, 1920), dtype=np.uint32)
t0 = time.perf_counter()
for i in range(frames):
    a = time.perf_counter()
    st = t.render(pod, 1920, 1080, seed=1, flags=flags, out=back)[2]
    b = time.perf_counter()
    print(f"launch {i}: from {a - t0:7.2f} s to {b - t0:7.2f} s  wall {1e3 * (b - a):8.1f} ms  kernel {st['render_ms']:8.1f} ms  ({st['kernel']}, {st['segments']} segments, upload {st['upload_ms']:.2f} ms)", flush=True)
    if pause:
        time.sleep(pause)
