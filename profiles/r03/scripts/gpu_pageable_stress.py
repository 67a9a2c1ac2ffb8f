"""Stress of the pageable read-back (no RT_HIP_FLAG_PERSISTENT_FRAME) under the allocation pattern of
tests/test_gpu_parity.py::test_noise_statistics_at_1080p_match_independent_generators, which once (1 run in 5) found float64
garbage in an array that no render ever wrote to: render into fresh numpy arrays, drop the RGBA one, make a float64 copy of
the float mean (a fresh 50 MB mapping that recycles freed pages), render again into fresh arrays, and check BOTH the new
frame and the bystander copy.  Prints what it finds; exits 1 on any corruption."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch  # noqa: F401
import rt_amd
from oracle import binding as oracle

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
w, h = 1920, 1080
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named("basic").set_sampling(2).describe(w, h)
want, want_rgb, _ = oracle.render(pod, w, h, seed=3)
bad_frames = bad_bystanders = 0
t0 = time.time()
for i in range(rounds):
    rgba, rgb, _ = t.render(pod, w, h, seed=3, want_rgb=True)
    ok1 = np.array_equal(rgba, want) and np.array_equal(rgb.view(np.uint32), want_rgb.view(np.uint32))
    del rgba
    bystander = rgb.astype(np.float64)  # fresh mapping, recycled pages
    rgba2, rgb2, _ = t.render(pod, w, h, seed=3, want_rgb=True)
    ok2 = np.array_equal(rgba2, want) and np.array_equal(rgb2.view(np.uint32), want_rgb.view(np.uint32))
    ok3 = np.array_equal(bystander, rgb.astype(np.float64))
    if not (ok1 and ok2):
        bad_frames += 1
        print(f"round {i}: frame wrong (first {ok1}, second {ok2})", flush=True)
    if not ok3:
        bad_bystanders += 1
        d = np.argwhere(bystander != rgb.astype(np.float64))
        print(f"round {i}: bystander array changed in {len(d)} places, first at {d[0].tolist()}, byte offset {int(((d[0][0] * w + d[0][1]) * 3 + d[0][2]) * 8)}", flush=True)
    del rgb, rgba2, rgb2, bystander
print(f"{rounds} rounds in {time.time() - t0:.1f} s: {bad_frames} wrong frames, {bad_bystanders} corrupted bystander arrays")
sys.exit(1 if bad_frames or bad_bystanders else 0)
