"""Kernel time of the scalar-register kernels for EVERY (spheres, planes) build, per library build:

    python tools/gpu_small_sweep.py <W> <H> <spp> <frames> lib.so [lib.so ...]      (libs under rt_amd/lib/)

The scene of a row: a ground sphere, `spheres - 1` small ones in front of the camera (a third of them metal), `planes` planes
(a floor just under the ground sphere's top, then walls behind and beside).  One process per library, min of <frames> kernel
times (HIP events).  Used to pick, per build, the waves-per-SIMD budget and the probe group size (kernels.hip): which build
spills how much moves with every change of the source (tools/kernel_registers.py)."""
import os, subprocess, sys
w, h, spp, frames = (int(v) for v in sys.argv[1:5])
code = f"""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, rt_amd
t = rt_amd.HipRayTracer(0)
frame = torch.empty(({h}, {w}), dtype=torch.int32, device='cuda:0')
s = torch.cuda.current_stream().cuda_stream
ivp = rt_amd.Scene.named('basic').describe({w}, {h}).inverse_view_projection[:]
materials = [(0, 1, 1, 1, 1, 0.5, 0.5), (0, 1, 0.2, 1, 1, 0.5, 0.5), (1, 1, 1, 1, 1, 0.05, 0.8)]
out = []
for planes in range(4):
    for spheres in range(1, 9 - planes):
        sp = [(0.0, -1000.0, 0.0, 1000.0, 0)] + [(-1.5 + 0.45 * i, 0.3 + 0.1 * (i % 3), -0.4 * (i % 2), 0.3, 1 + i % 2) for i in range(spheres - 1)]
        pl = [(0.0, 1.0, 0.0, 0.001, 0), (0.0, 0.0, 1.0, 4.0, 1), (1.0, 0.0, 0.0, 5.0, 0)][:planes]
        pod = rt_amd.scene_from_arrays(sp, pl, materials, samples_per_pixel={spp}, max_bounces=10, inverse_view_projection=ivp)
        t.upload(pod)
        def dev():
            t.render_device({w}, {h}, frame.data_ptr(), seed=1, stream=s); return t.stats()['render_ms']
        dev(); dev()
        out.append((spheres, planes, min(dev() for _ in range({frames}))))
print(' '.join('%d+%d:%.4f' % o for o in out))
"""
table = {}
for lib in sys.argv[5:]:
    env = dict(os.environ, RT_HIP_LIBRARY=os.path.abspath(f"rt_amd/lib/{lib}"))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if res.returncode != 0:
        print(lib, res.stderr[-600:], flush=True)
        continue
    for item in res.stdout.split():
        key, ms = item.split(":")
        table.setdefault(key, {})[lib] = float(ms)
libs = sys.argv[5:]
print(f"{'spheres+planes':14s} " + " ".join(f"{l.replace('librt_hip', '').replace('.so', '') or '(product)':>12s}" for l in libs))
for key, row in table.items():
    best = min(row.values())
    print(f"{key:14s} " + " ".join(f"{row.get(l, float('nan')):11.4f}{'*' if row.get(l) == best else ' '}" for l in libs), flush=True)
