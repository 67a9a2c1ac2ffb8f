"""How far RT_HIP_FLAG_FAST frames are from the parity-contract frames (the numbers tests/test_gpu_fast.py's bounds come from)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch  # noqa
import rt_amd
from rt_amd import capi

t = rt_amd.HipRayTracer(0)
CASES = [("synthetic-100k", 1920, 1080, 64)] if len(sys.argv) > 1 and sys.argv[1] == "config5" else [("basic", 1920, 1080, 256), ("basic", 1920, 1080, 64), ("dielectric", 1920, 1080, 256), ("basic", 3840, 2160, 64), ("synthetic-2000", 480, 270, 16), ("synthetic-100k", 240, 135, 4), ("basic", 256, 256, 1)]
for name, w, h, spp in CASES:
    pod = rt_amd.Scene.named(name).set_sampling(spp).describe(w, h)
    a8, a, sa = t.render(pod, w, h, seed=1, want_rgb=True)
    b8, b, sb = t.render(pod, w, h, seed=1, flags=capi.RT_HIP_FLAG_FAST, want_rgb=True)
    rel = np.abs(a - b).max(axis=2) / np.maximum(np.abs(a).max(axis=2), 1e-6)
    ua = np.stack([(a8 >> s) & 255 for s in (24, 16, 8)], -1).astype(int)
    ub = np.stack([(b8 >> s) & 255 for s in (24, 16, 8)], -1).astype(int)
    d8 = np.abs(ua - ub).max(axis=2)
    q = np.quantile(rel, [0.5, 0.9, 0.99, 0.999, 0.9999])
    print(f"{name} {w}x{h}x{spp}: exact {sa['render_ms']:.3f} ms, fast {sb['render_ms']:.3f} ms ({sa['render_ms'] / sb['render_ms']:.3f}x) kernel {sb['kernel']}")
    print(f"   rel err of the float mean: median {q[0]:.2e} p90 {q[1]:.2e} p99 {q[2]:.2e} p99.9 {q[3]:.2e} p99.99 {q[4]:.2e} max {rel.max():.2e}")
    print(f"   pixels with rel err > 2e-5: {(rel > 2e-5).mean() * 100:.4f} %   > 1e-3: {(rel > 1e-3).mean() * 100:.4f} %   > 1/spp: {(rel > 1.0 / spp).mean() * 100:.4f} %")
    print(f"   RGBA8: identical {(d8 == 0).mean() * 100:.3f} %, off by 1 {(d8 == 1).mean() * 100:.3f} %, off by more {(d8 > 1).mean() * 100:.4f} % (max {d8.max()})")
    print(f"   segments exact {sa['segments']} fast {sb['segments']} ({(sb['segments'] - sa['segments']) / sa['segments']:+.2e})   frame mean exact {a.mean():.6f} fast {b.mean():.6f}")
