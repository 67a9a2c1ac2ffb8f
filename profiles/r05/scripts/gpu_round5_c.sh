#!/bin/bash
# Round 5, visit C: register budget and code-generation A/B of the contract-v4 kernels — the scatter function packed into the LDS
# geometry word (71 registers, 7 waves) against a word of its own (59, 8 waves); SLP vectorisation off (no v_pk_*); the 1..4-sphere
# kernels compiled for 8 waves per SIMD.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{
for cfg in "basic 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256" "dielectric 1920 1080 256"; do
  echo "== $cfg =="
  timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so librt_hip_nopack.so librt_hip_noslp.so librt_hip_w8.so || exit 1
done
echo "== basic 1920 1080 256, tilted camera =="
AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py basic 1920 1080 256 15 librt_hip.so librt_hip_nopack.so librt_hip_noslp.so librt_hip_w8.so || exit 1
} 2>&1 | tee gpurun_out/r05/c_codegen_ab.txt
