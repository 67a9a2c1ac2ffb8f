#!/bin/bash
# Round 5, visit B: arithmetic contract v4 (one generator step per random<T>() call, primary rays from a per-pixel base, one
# division in the general camera form) — parity suite against the v4 oracle, then A/B: round 4, phase 1, contract v4.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/b_pytest_gpu.txt 2>&1; rc=$?; tail -5 gpurun_out/r05/b_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
{
for cfg in "basic 1920 1080 256" "basic 1920 1080 64" "dielectric 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256"; do
  echo "== $cfg =="
  timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip_r4.so librt_hip_p1.so librt_hip.so || exit 1
done
echo "== basic 1920 1080 256, tilted camera =="
AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py basic 1920 1080 256 15 librt_hip_r4.so librt_hip_p1.so librt_hip.so || exit 1
echo "== scenes/basic_plane.toml 1920 1080 256, tilted camera =="
AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py scenes/basic_plane.toml 1920 1080 256 15 librt_hip_r4.so librt_hip_p1.so librt_hip.so || exit 1
} 2>&1 | tee gpurun_out/r05/b_contract_v4_ab.txt
