#!/bin/bash
# Round 5, visit A: phase 1 of the small kernel's diet — bit-preserving cuts (no multiply / select per trip for the stream
# position, no vote in front of the tail, sample 0 behind a vote, the bounce count touched by bounces only, absorbed /
# out-of-bounces inside the scatter block) — parity suite against the UNCHANGED oracle, then A/B against round 4's build.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/a_pytest_gpu.txt 2>&1; rc=$?; tail -5 gpurun_out/r05/a_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
for cfg in "basic 1920 1080 256" "basic 1920 1080 64" "dielectric 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256"; do
  echo "== $cfg =="
  timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip_r4.so librt_hip.so || exit 1
done 2>&1 | tee gpurun_out/r05/a_phase1_ab.txt
