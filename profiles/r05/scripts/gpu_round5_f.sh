#!/bin/bash
# Round 5, visit F: (1) the new resident-with-rolling-items parity cases; (2) every (spheres, planes) build of the scalar-register
# kernels at 7 and at 6 waves per SIMD (group size 3 from five spheres on); (3) the LDS-resident kernel with a tile per wave
# against rolling items, by scene size.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest (resident rolling, headline, launch paths) =="
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_half_chunks.py tests/test_gpu_launch_paths.py -m gpu -q -x --timeout 600 > gpurun_out/r05/f_pytest_gpu.txt 2>&1; rc=$?; tail -6 gpurun_out/r05/f_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
echo "== small kernels, every build: 7 waves (product) / 6 waves everywhere / 7 waves for 7 + 1 too =="
timeout -k 10 600 python tools/gpu_small_sweep.py 1920 1080 256 8 librt_hip.so librt_hip_w6.so librt_hip_w7.so 2>&1 | tee gpurun_out/r05/f_small_sweep.txt
{
for cfg in "synthetic-12 1920 1080 64" "synthetic-24 1920 1080 64" "synthetic-64 1920 1080 64" "synthetic-64 1920 1080 256" "synthetic-200 1920 1080 64" "synthetic-700 1920 1080 64"; do
  echo "== $cfg =="
  timeout -k 10 400 python tools/gpu_ab.py $cfg 8 librt_hip_noroll.so librt_hip.so librt_hip_roll9.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/f_resident_rolling_ab.txt
