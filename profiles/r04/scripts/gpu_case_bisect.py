"""Which round-4 change broke a random-scene case?  Runs case(s) of tests/test_gpu_parity.py::test_random_scenes_are_bit_exact against
every build named on the command line (one process per build: RT_HIP_LIBRARY) and prints, per flag set, how many pixels differ
from the oracle and where the first ones are.   usage: python tools/gpu_case_bisect.py <case> lib.so [lib.so ...]"""
import os
import subprocess
import sys

case = int(sys.argv[1])
code = f"""
import sys; sys.path.insert(0, '.')
import numpy as np
import rt_amd
from rt_amd import capi
from oracle import binding as oracle
from tests.test_gpu_parity import random_scene, FORCE_RESIDENT, FORCE_TILED, FORCE_STREAMED, SM
rng = np.random.default_rng(1000 + {case})
spheres, planes, materials, camera = random_scene(rng)
width, height = int(rng.integers(17, 140)), int(rng.integers(9, 90))
spp, bounces = int(rng.integers(1, 40)), int(rng.integers(1, 12))
if {case} % 4 == 3:
    spp = int(rng.integers(40, 140))
ivp = camera.describe(width, height).inverse_view_projection[:]
pod = rt_amd.scene_from_arrays(spheres, planes, materials, samples_per_pixel=spp, max_bounces=bounces, inverse_view_projection=ivp)
seed = int(rng.integers(0, 2**63))
print('scene', len(spheres), 'spheres', len(planes), 'planes', width, 'x', height, 'spp', spp, 'bounces', bounces)
HALF, WHOLE = capi.RT_HIP_FLAG_FORCE_HALF_CHUNKS, capi.RT_HIP_FLAG_FORCE_WHOLE_CHUNKS
t = rt_amd.HipRayTracer(0)
wanted = {{}}
for flags in (0, FORCE_RESIDENT, FORCE_TILED, FORCE_STREAMED, SM, SM | FORCE_STREAMED, SM | FORCE_TILED, HALF, WHOLE, HALF | FORCE_RESIDENT, HALF | FORCE_STREAMED, WHOLE | FORCE_TILED, SM | FORCE_STREAMED):
    got_rgba, got_rgb, stats = t.render(pod, width, height, seed=seed, flags=flags, want_rgb=True)
    sm = bool(flags & SM)
    if sm not in wanted:
        wanted[sm] = oracle.render(pod, width, height, seed=seed, sm_materials=sm)
    want_rgba, want_rgb, want_stats = wanted[sm]
    same = (got_rgb.view(np.uint32) == want_rgb.view(np.uint32)) | (np.isnan(got_rgb) & np.isnan(want_rgb))
    bad = np.argwhere(~same.all(axis=-1))
    note = ''
    if len(bad):
        y, x = bad[0]
        note = f' first (y, x) = ({{y}}, {{x}}): got {{got_rgb[y, x]}} want {{want_rgb[y, x]}}; rows {{sorted(set(bad[:, 0].tolist()))[:12]}}'
    print(f'flags {{flags:4d}} {{stats["kernel"]:9s}} segments {{stats["segments"]}} vs {{want_stats["segments"]}}: {{len(bad)}} pixels differ' + note)
"""
for lib in sys.argv[2:]:
    env = dict(os.environ, RT_HIP_LIBRARY=os.path.abspath(f"rt_amd/lib/{lib}"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"==== {lib} ====\n{out.stdout}{out.stderr[-1500:] if out.returncode else ''}", flush=True)
