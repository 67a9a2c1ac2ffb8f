"""Dynamic side of the per-region ISA table: how often each part of a render_queue loop trip runs, and with how many lanes.

    RT_HIP_LIBRARY=rt_amd/lib/librt_hip_regions.so python tools/region_profile.py [scene W H spp]

Needs the instrumented experiment build (`make variant NAME=regions DEFS=-DRT_HIP_REGION_COUNTERS=1`): it counts, per
wave and region, executions and active lanes in scalar registers and adds them up at the end of the launch."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import torch  # noqa: F401  (load order: torch's HIP runtime first)

import rt_amd
from rt_amd import capi

NAMES = ["trip", "query: probes", "sqrt half of a sphere", "hit: lookups + normal", "miss: sky + end of sample", "metal: normalise dir + reflect", "hand-out vote",
         "take item", "tail: two draws", "scatter: 3rd draw, unit vector", "restart: primary ray", "normalise new direction", "absorbed / out of bounces"]
scene, w, h, spp = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("basic", 1920, 1080, 256)
t = rt_amd.HipRayTracer(0)
lib = capi.hip_lib()
pod = (rt_amd.Scene.load(scene) if scene.endswith(".toml") else rt_amd.Scene.named(scene)).set_sampling(spp).describe(w, h)
t.upload(pod)
frame = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
t.render_device(w, h, frame.data_ptr(), seed=1, stream=torch.cuda.current_stream().cuda_stream)
stats = t.stats()
out = (C.c_uint64 * 26)()
fn = lib.rt_hip_debug_region_counters
fn.argtypes = [C.c_void_p, C.c_void_p]
assert fn(t._ctx, out) == 0
runs, lanes = list(out[:13]), list(out[13:])
samples = w * h * spp
print(f"{scene} {w}x{h}x{spp}: {samples} samples, {stats['segments']} segments, kernel {stats['kernel']}, {stats['render_ms']:.3f} ms (instrumented)")
print(f"{'region':34s} {'runs':>12s} {'runs/trip':>10s} {'lanes/run':>10s} {'lane-runs/sample':>17s}")
for name, r, l in zip(NAMES, runs, lanes):
    print(f"{name:34s} {r:12d} {r / max(runs[0], 1):10.3f} {l / max(r, 1):10.2f} {l / samples:17.4f}")
print(f"trips per 64 samples: {runs[0] * 64 / samples:.2f}   (ideal = segments per sample = {stats['segments'] / samples:.3f} with every lane holding a ray in every trip)")
