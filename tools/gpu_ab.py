"""A/B timing of builds of librt_hip.so, one process per build, interleaved over two rounds:

    python tools/gpu_ab.py <scene name | file.toml> <W> <H> <spp> <reps> [lib.so ...]      (libs under rt_amd/lib/)

Per build: the kernel alone (scene resident, frame in HBM: rt_hip_stats.render_ms) and the drop-in call (host scene in,
frame in a page-locked host back buffer out: wall clock and the kernel inside it).  min / median over <reps> frames."""
import os, subprocess, sys
scene, w, h, spp, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
code = f"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import rt_amd, os
from rt_amd import capi
if '_r3' in os.environ.get('RT_HIP_LIBRARY', ''):  # a round-3 build: bind what its ABI had
    capi.RT_HIP_SYMBOLS = [e for e in capi.RT_HIP_SYMBOLS if e[0] != 'rt_hip_live_frame_locks']
flags = int('{os.environ.get("AB_FLAGS", "0")}')
t = rt_amd.HipRayTracer(0)
scene_ = (rt_amd.Scene.load('{scene}') if '{scene}'.endswith('.toml') else rt_amd.Scene.named('{scene}')).set_sampling({spp})
if os.environ.get('AB_TILT'):  # bench.py --tilt: a camera that is not axis-aligned (w varies over the frame: the general-camera kernels)
    scene_.set_camera((0.2, 1.2, 3.0), (0.0, -0.15, -1.0))
pod = scene_.describe({w}, {h})
t.upload(pod)
frame = torch.empty(({h}, {w}), dtype=torch.int32, device='cuda:0')
s = torch.cuda.current_stream().cuda_stream
def dev():
    t.render_device({w}, {h}, frame.data_ptr(), seed=1, flags=flags, stream=s); return t.stats()['render_ms']
dev(); dev()
k = sorted(dev() for _ in range({reps}))
back = np.zeros(({h}, {w}), dtype=np.uint32)
def drop():
    t0 = time.perf_counter(); st = t.render(pod, {w}, {h}, seed=1, flags=flags | capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)[2]; return (time.perf_counter() - t0) * 1e3, st['render_ms']
drop(); drop()
d = [drop() for _ in range({reps})]
wall = sorted(x[0] for x in d); kin = sorted(x[1] for x in d)
m = len(k) // 2
print('kernel %.4f / %.4f   drop-in wall %.4f / %.4f (kernel inside %.4f / %.4f)' % (k[0], k[m], wall[0], wall[m], kin[0], kin[m]))
"""
for rnd in range(2):
    for lib in (sys.argv[6:] or ["librt_hip_old.so", "librt_hip.so"]):
        env = dict(os.environ, RT_HIP_LIBRARY=os.path.abspath(f"rt_amd/lib/{lib}"))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"{lib:32s}", out.stdout.strip() or out.stderr[-400:], flush=True)
