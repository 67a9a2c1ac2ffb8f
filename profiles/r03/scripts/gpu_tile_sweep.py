"""Whole frames through the drop-in call (back buffer placed on the GPU's node and page-locked by the library), tile size
and shape from the environment (experiment build librt_hip_knobs.so):
    RT_HIP_LIBRARY=rt_amd/lib/librt_hip_knobs.so python tools/gpu_tile_sweep.py
The candidates of a case are measured INTERLEAVED (6 rounds x 8 frames each, after a warm-up round that is thrown away):
kernel ms inside the call, median over all rounds, and the spread of the rounds' medians."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import rt_amd
from rt_amd import capi

P = capi.RT_HIP_FLAG_PERSISTENT_FRAME
t = rt_amd.HipRayTracer(0)
CASES = [("basic", 1920, 1080, 256), ("dielectric", 1920, 1080, 256), ("basic", 3840, 2160, 256), ("basic", 1280, 720, 256), ("basic", 1920, 1080, 64), ("basic", 1920, 1080, 128), ("basic", 1920, 1080, 16)]


def set_shape(p, tw):
    for key, val in (("RT_HIP_TILE_LOG2", p), ("RT_HIP_TILE_W_LOG2", tw)):
        if val is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = str(val)


for scene, w, h, spp in CASES:
    pod = rt_amd.Scene.named(scene).set_sampling(spp).describe(w, h)
    frame = np.zeros((h, w), dtype=np.uint32)
    chunks = (spp + 15) // 16
    shapes = [(None, None)]
    for p in range(2, 8):
        if 64 <= (chunks << p) <= 256:
            for tw in sorted({(p + 1) // 2, min(p, 3), min(p, 4)}):
                shapes.append((p, tw))
    frames = 4 if w > 2000 else 8
    samples = {s: [] for s in shapes}
    for rnd in range(7):
        for s in shapes:
            set_shape(*s)
            t.render(pod, w, h, flags=P, out=frame)
            ms = [t.render(pod, w, h, flags=P, out=frame)[2]["render_ms"] for _ in range(frames)]
            if rnd:
                samples[s].append(ms)
    print(f"--- {scene} {w}x{h} at {spp} spp", flush=True)
    for (p, tw), rounds in samples.items():
        medians = [float(np.median(r)) for r in rounds]
        label = "as chosen by the library" if p is None else f"{1 << p:3d} pixels as {1 << tw:2d} x {(1 << p) >> tw}"
        print(f"  {label:28s} kernel {np.median(np.concatenate(rounds)):.4f} ms   (rounds {min(medians):.4f} .. {max(medians):.4f})", flush=True)
    t.forget_frame()
t.close()
