#!/bin/bash
# Round 4, visit P: 6 against 7 waves per SIMD for the scalar-register kernels whose eight primitive slots are all taken and
# include planes (7 + 1 measured before: dielectric + plane; here 6 + 2 and 5 + 3: the dielectric scene with walls for spheres).
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
python - <<'PY'
import re
text = open('scenes/dielectric_plane.toml').read()
blocks = text.split('[[spheres]]')
head, spheres = blocks[0], blocks[1:]
wall = lambda n, d, m: f"[[planes]]\nnormal = [{n[0]}, {n[1]}, {n[2]}]\nposition = [{-n[0]*d}, {-n[1]*d}, {-n[2]*d}]\nmaterial = {m}\n\n"
for name, keep, walls in (("six_two", 6, [((0.0, 0.0, 1.0), 6.0, 2)]), ("five_three", 5, [((0.0, 0.0, 1.0), 6.0, 2), ((1.0, 0.0, 0.0), 6.0, 1)])):
    out = head + "".join(wall(*w) for w in walls) + "".join('[[spheres]]' + s for s in spheres[:keep])
    open(f'/tmp/{name}.toml', 'w').write(out)
import sys; sys.path.insert(0, '.')
import rt_amd
for name in ("six_two", "five_three"):
    pod = rt_amd.Scene.load(f'/tmp/{name}.toml').describe(64, 36)
    print(name, pod.n_spheres, 'spheres', pod.n_planes, 'planes')
PY
for scene in /tmp/six_two.toml /tmp/five_three.toml dielectric_plane; do
  echo "== $scene 1920x1080x256 =="
  timeout -k 10 500 python tools/gpu_ab.py $scene 1920 1080 256 15 librt_hip_w7.so librt_hip.so || exit 1
done 2>&1 | tee gpurun_out/r04/waves_full_ab_planes.txt
