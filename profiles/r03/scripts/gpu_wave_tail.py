"""How a persistent big-scene launch ends: per-wave start / queue-dry / retire times (instrumented experiment build).

    make variant NAME=clocks DEFS=-DRT_HIP_WAVE_CLOCKS=1
    RT_HIP_LIBRARY=rt_amd/lib/librt_hip_clocks.so python tools/gpu_wave_tail.py [scene W H spp [launches]]

Prints, per launch: kernel time, when the tile queue ran dry for the first and the last wave, when waves retired
(quantiles), and the share of wave-time lost between a wave's retirement and the end of the launch."""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401

import rt_amd
from rt_amd import capi

scene, w, h, spp = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("synthetic-100k", 1920, 1080, 64)
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 2
t = rt_amd.HipRayTracer(0)
lib = capi.hip_lib()
fn = lib.rt_hip_debug_wave_clocks
fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
pod = rt_amd.Scene.named(scene).set_sampling(spp).describe(w, h)
t.upload(pod)
frame = torch.empty((h, w), dtype=torch.int32, device="cuda:0")
for launch in range(launches):
    t.render_device(w, h, frame.data_ptr(), seed=1, stream=torch.cuda.current_stream().cuda_stream)
    stats = t.stats()
    n = C.c_uint32(16384)
    out = np.zeros((16384, 3), dtype=np.uint64)
    assert fn(t._ctx, out.ctypes.data, C.byref(n)) == 0
    clocks = out[: n.value]
    ran = clocks[:, 2] > 0
    start, dry, end = (clocks[ran, k].astype(np.float64) for k in range(3))
    t0 = start.min()
    tick_ms = 1e-5  # 100 MHz
    total = (end.max() - t0) * tick_ms
    worked = dry > 0  # waves that pulled at least one tile and later found the queue dry (surplus workgroups start dry)
    retire = (end - t0) * tick_ms
    late = start > t0 + 0.5 * (end.max() - t0)  # surplus workgroups: scheduled when the first ones retired
    print(f"launch {launch}: {scene} {w}x{h}x{spp}, {stats['kernel']} kernel {stats['render_ms']:.1f} ms; {int(ran.sum())} waves recorded, {int(late.sum())} of them started in the second half (surplus workgroups)")
    r = retire[~late]
    d = ((dry[~late & worked] - t0) * tick_ms) if (~late & worked).any() else np.zeros(1)
    print(f"  queue found dry: first wave at {d.min():.1f} ms, median {np.median(d):.1f}, last {d.max():.1f}")
    qs = [0, 1, 5, 25, 50, 75, 95, 99, 100]
    print("  waves retired (ms):   " + "  ".join(f"p{q}={np.percentile(r, q):.0f}" for q in qs))
    lost = (total - r).sum() / (total * len(r))
    print(f"  wave-time between a wave's retirement and the end of the launch: {100 * lost:.1f} % of the launch ({total:.1f} ms x {len(r)} waves)")
    active = [(r > x).mean() for x in np.linspace(0, total, 21)]
    print("  share of waves still running at 0 %, 5 %, ... 100 % of the launch: " + " ".join(f"{a:.2f}" for a in active))
