#!/bin/bash
# Round 5, visit X: with the streamed kernel's dense build 9 % faster, where does the LDS-resident kernel's lead end now?  The launch
# code's choice (resident up to 3000 primitives) against the streamed kernel forced (AB_FLAGS=32), 1080p x 64 spp, kernel ms.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{
for n in 400 700 1000 1100 1500 2000 3000; do
  echo "== synthetic-$n 1920 1080 64: the launch code's choice (resident) =="; timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 4 librt_hip.so || exit 1
  echo "== synthetic-$n 1920 1080 64: streamed forced =="; AB_FLAGS=32 timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 4 librt_hip.so || exit 1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/x_resident_vs_dense_streamed.txt
