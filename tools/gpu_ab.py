"""A/B timing of builds of librt_hip.so, one process per build, interleaved:
python tools/gpu_ab.py <scene> <W> <H> <spp> <reps> [lib.so ...]   (default: librt_hip_old.so librt_hip.so, under rt_amd/lib/)"""
import os, subprocess, sys
scene, w, h, spp, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
code = f"""
import sys; sys.path.insert(0, '.')
import rt_amd
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named('{scene}').set_sampling({spp}).describe({w}, {h})
t.render(pod, {w}, {h}, seed=1)
ms = [t.render(pod, {w}, {h}, seed=1)[2]['render_ms'] for _ in range({reps})]
print(min(ms), sorted(ms)[len(ms)//2])
"""
for rnd in range(2):
    for lib in (sys.argv[6:] or ["librt_hip_old.so", "librt_hip.so"]):
        env = dict(os.environ, RT_HIP_LIBRARY=os.path.abspath(f"rt_amd/lib/{lib}"))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(lib, out.stdout.strip() or out.stderr[-300:], flush=True)
