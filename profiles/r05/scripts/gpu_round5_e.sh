#!/bin/bash
# Round 5, visit E: the whole GPU suite with this round's new cases (whole-frame digests, the drop-in call at full size, real
# soagen columns on the GPU, the early line / budget of the N > 1 flow), bench.py's line, probe-group / wave-budget A/B for the
# 5..8-sphere kernels, and the rocprofv3 passes of the headline.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/e_pytest_gpu.txt 2>&1; rc=$?; tail -8 gpurun_out/r05/e_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
echo "== bench.py =="
timeout -k 10 300 python bench.py > gpurun_out/r05/e_bench.jsonl 2> gpurun_out/r05/e_bench.err; rc=$?; cut -c1-400 gpurun_out/r05/e_bench.jsonl; tail -3 gpurun_out/r05/e_bench.err
[ $rc -ne 0 ] && exit $rc
{
for cfg in "dielectric 1920 1080 256" "dielectric_plane 1920 1080 256" "synthetic-8 1920 1080 256" "synthetic-5 1920 1080 256"; do
  echo "== $cfg =="
  timeout -k 10 400 python tools/gpu_ab.py $cfg 12 librt_hip.so librt_hip_g3.so librt_hip_g8.so librt_hip_many6.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/e_groups_ab.txt
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
