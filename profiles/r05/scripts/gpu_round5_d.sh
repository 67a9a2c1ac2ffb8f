#!/bin/bash
# Round 5, visit D: contract v4 as it will ship (plane distance by reciprocal, plane hits skip the sphere normal, votes combined on
# the scalar unit, no SLP vectorisation, 5..8 spheres probed in groups of four) — parity suite, then A/B against round 4 and
# between probe-group sizes / wave budgets for the 5..8-sphere kernels.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/d_pytest_gpu.txt 2>&1; rc=$?; tail -5 gpurun_out/r05/d_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
{
for cfg in "basic 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256" "synthetic-64 1920 1080 256"; do
  echo "== $cfg =="
  timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip_r4.so librt_hip.so || exit 1
done
echo "== basic 1920 1080 256, tilted camera =="
AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py basic 1920 1080 256 15 librt_hip_r4.so librt_hip.so || exit 1
echo "== scenes/basic_plane.toml 1920 1080 256, tilted camera =="
AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py scenes/basic_plane.toml 1920 1080 256 15 librt_hip_r4.so librt_hip.so || exit 1
for cfg in "dielectric 1920 1080 256" "dielectric_plane 1920 1080 256" "synthetic-8 1920 1080 256" "synthetic-5 1920 1080 256"; do
  echo "== $cfg =="
  timeout -k 10 400 python tools/gpu_ab.py $cfg 12 librt_hip_r4.so librt_hip.so librt_hip_g3.so librt_hip_g8.so librt_hip_many6.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/d_ab.txt
