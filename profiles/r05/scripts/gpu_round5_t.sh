#!/bin/bash
# Round 5, visit T: how far does the tile-per-wave hand-out stay ahead of the rolling items?  Visit R: 1024 spheres 0.77 of the peak
# in the LDS-resident kernel, 1100 spheres 0.70 in the streamed one — and from 40 spheres on the resident kernel does not read its LDS
# copy of the spheres at all.  librt_hip_bigres.so: the resident kernel stages only the planes then and takes any number of spheres
# (RT_HIP_FLAG_FORCE_RESIDENT = AB_FLAGS=2; the shipped library ignores the flag beyond 1024 primitives and runs the streamed kernel).
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== parity of the experiment build: 2000 and 1500 spheres, resident forced, against the oracle =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_bigres.so timeout -k 10 300 python3 - <<'PY' || exit 1
import sys; sys.path.insert(0, '.')
import numpy as np
import rt_amd
from rt_amd import capi
from oracle import binding as oracle
t = rt_amd.HipRayTracer(0)
for name, w, h, spp in (("synthetic-2000", 96, 54, 8), ("synthetic-1500", 64, 36, 20)):
    pod = rt_amd.Scene.named(name).set_sampling(spp).describe(w, h)
    for flags in (capi.RT_HIP_FLAG_FORCE_RESIDENT, capi.RT_HIP_FLAG_FORCE_RESIDENT | capi.RT_HIP_FLAG_FORCE_HALF_CHUNKS):
        got, got_rgb, stats = t.render(pod, w, h, seed=3, flags=flags, want_rgb=True)
        want, want_rgb, want_stats = oracle.render(pod, w, h, seed=3)
        ok = np.array_equal(got, want) and np.array_equal(got_rgb.view(np.uint32), want_rgb.view(np.uint32)) and stats["segments"] == want_stats["segments"]
        print(name, flags, stats["kernel"], "bit-exact" if ok else "MISMATCH")
        assert ok and stats["kernel"] == "resident"
PY
{
for cfg in "synthetic-1000 1920 1080 64 4" "synthetic-1100 1920 1080 64 4" "synthetic-2000 1920 1080 64 4" "synthetic-5000 1920 1080 64 3" "synthetic-10000 1920 1080 64 2" "synthetic-100k 1920 1080 64 1"; do
  echo "== $cfg, resident forced =="; AB_FLAGS=2 timeout -k 10 500 python tools/gpu_ab.py $cfg librt_hip.so librt_hip_bigres.so || exit 1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/t_resident_beyond_1024_ab.txt
