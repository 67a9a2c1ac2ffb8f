"""Round 4 diagnosis of VERDICT r3 weak #2 (a test's own float64 array found full of RGBA8 words): what does the HIP runtime
do with a PAGEABLE destination of hipMemcpyAsync — the only way the round-3 module handed caller memory to it?

Run with AMD_LOG_LEVEL=4 AMD_LOG_MASK=1792 (decimal: 0x700 was read as 0) (copy paths + resources) against the ROUND-3 library: the log says, per copy,
whether the runtime staged it through its own buffer or page-locked the caller's memory ("Pinned resource"), and the
allocation log below says where numpy put rgba / rgb / the float64 copy, call after call, in the test's own sequence."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import rt_amd  # noqa: E402
from rt_amd import capi  # noqa: E402

# (the round-3 library lacks what round 4 added to the ABI: bind what it has)
capi.RT_HIP_SYMBOLS = [entry for entry in capi.RT_HIP_SYMBOLS if entry[0] != "rt_hip_live_frame_locks"]

width, height = 1920, 1080
scene = rt_amd.Scene.named("basic")
tracer = rt_amd.HipRayTracer(device=0)


def traced(spp, seed, tag):
    pod = scene.set_sampling(spp).describe(width, height)
    rgba, rgb, _ = tracer.render(pod, width, height, seed=seed, want_rgb=True)
    a, b = rgba.ctypes.data, rgb.ctypes.data
    del rgba  # what `render(...)[1]` does: the tuple and the packed frame go first
    f64 = rgb.astype(np.float64)
    print(f"{tag}: rgba {a:#x}..{a + 4 * width * height:#x}  rgb {b:#x}..{b + 12 * width * height:#x}  float64 {f64.ctypes.data:#x}..{f64.ctypes.data + f64.nbytes:#x}", flush=True)
    return f64


# what the suite does right before the test that failed: config 4's frame, 3840 x 2160 with the float mean (99.5 MB: the one
# copy of the whole suite above 32 MiB) into fresh numpy arrays, dropped afterwards
big = scene.set_sampling(1).describe(3840, 2160)
rgba, rgb, _ = tracer.render(big, 3840, 2160, seed=1, want_rgb=True)
print(f"4K: rgba {rgba.ctypes.data:#x}..{rgba.ctypes.data + rgba.nbytes:#x}  rgb {rgb.ctypes.data:#x}..{rgb.ctypes.data + rgb.nbytes:#x}", flush=True)
del rgba, rgb

for round_ in range(4):
    truth = traced(8, 99, f"round {round_} truth")
    ours = traced(4, 5, f"round {round_} ours ")
    for name, img in (("truth", truth), ("ours", ours)):
        wild = np.argwhere(~(np.abs(img) < 4.0).all(axis=-1))
        if len(wild):
            y, x = wild[0]
            print(f"round {round_}: {name} has {len(wild)} wild pixels, first at (y, x) = ({y}, {x}), address {img.ctypes.data + (y * width + x) * 24:#x}, bytes {img[y, x].tobytes().hex()}", flush=True)
    del truth, ours
print("done")
