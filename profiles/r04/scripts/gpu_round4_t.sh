#!/bin/bash
# Round 4, visit T: the scalar-register kernels of 5-8 spheres probing their spheres in groups of four (fewer discriminants alive:
# 65-67 registers, no spills) against all at once (72 registers, 12-16 bytes of scratch from seven spheres up).
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
for scene in dielectric dielectric_plane synthetic-8 synthetic-6 synthetic-5; do
  echo "== $scene 1920x1080x256 =="
  timeout -k 10 500 python tools/gpu_ab.py $scene 1920 1080 256 15 librt_hip.so librt_hip_g4.so librt_hip_g4w7.so librt_hip_g4w8.so || exit 1
done 2>&1 | tee gpurun_out/r04/probe_groups_ab.txt
