#!/bin/bash
# Round 5, visit V: the streamed kernel's occupancy.  It is compiled for 5 waves per SIMD because the cooperative scan of sparse waves
# (four 16-byte loads in flight per lane) takes 95 registers in the half-chunk build; without that scan the same loop needs 69 (80 in
# the half-chunk build).  In a frame that fills the device only the launch's last waves are sparse.  Builds without the cooperative
# scan at 5, 6 and 7 waves per SIMD (and as many workgroups per CU) against the shipped kernel: config 5 and smaller fields, kernel ms.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{
for cfg in "synthetic-3100 1920 1080 64 3" "synthetic-10000 1920 1080 64 2" "synthetic-100k 1920 1080 64 1"; do
  echo "== $cfg =="; timeout -k 10 900 python tools/gpu_ab.py $cfg librt_hip.so librt_hip_big5n.so librt_hip_big6n.so librt_hip_big7n.so || exit 1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/v_streamed_occupancy_ab.txt
