"""Where do half-chunk items pay?  Whole frames and shares through the drop-in call (host frame) with the launch code's own
choice, with half chunks forced and with whole chunks forced; kernel ms (median of 30)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import rt_amd
from rt_amd import capi

t = rt_amd.HipRayTracer(0)
P = capi.RT_HIP_FLAG_PERSISTENT_FRAME
for w, h, spp in ((1920, 1080, 16), (1920, 1080, 24), (1920, 1080, 32), (1920, 1080, 48), (1920, 1080, 64), (1280, 720, 64), (1280, 720, 128), (800, 600, 64), (800, 600, 256), (640, 360, 256), (3840, 2160, 16)):
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(w, h)
    frame = np.zeros((h, w), dtype=np.uint32)
    line = [f"{w}x{h} at {spp:3d} spp ({w * h * ((spp + 15) // 16) / 524288:.1f} chunks per lane):"]
    res = {}
    for rnd in range(3):
        for name, flags in (("auto", 0), ("half", capi.RT_HIP_FLAG_FORCE_HALF_CHUNKS), ("whole", capi.RT_HIP_FLAG_FORCE_WHOLE_CHUNKS)):
            for _ in range(5):
                t.render(pod, w, h, flags=P | flags, out=frame)
            ms = [t.render(pod, w, h, flags=P | flags, out=frame)[2]["render_ms"] for _ in range(10)]
            res.setdefault(name, []).extend(ms)
    for name in ("auto", "half", "whole"):
        line.append(f"{name} {np.median(res[name]):.4f}")
    print("   ".join(line), flush=True)
    t.forget_frame()
t.close()
