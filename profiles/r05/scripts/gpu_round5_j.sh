#!/bin/bash
# Round 5, visit J: the plane test without its per-lane "in reach" (one-comparison guard on the reciprocal, select_hit on masks) against
# visit I's kernels and against a build without the "hopeless" vote; the whole GPU suite; a parity soak over random scenes; one
# bench.py line per BASELINE.json configuration.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/j_pytest_gpu.txt 2>&1; rc=$?; tail -8 gpurun_out/r05/j_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
{
for cfg in "scenes/basic_plane.toml 1920 1080 256" "dielectric_plane 1920 1080 256" "basic 1920 1080 256"; do
  echo "== $cfg =="; timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so librt_hip_visit_i.so librt_hip_always_rcp.so || exit 1
done
echo "== scenes/basic_plane.toml 1920 1080 256, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py scenes/basic_plane.toml 1920 1080 256 15 librt_hip.so librt_hip_visit_i.so librt_hip_always_rcp.so || exit 1
} 2>&1 | tee gpurun_out/r05/j_plane_ab.txt
echo "== soak: random scenes =="
RT_HIP_RANDOM_CASES=1500 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k random_scenes > gpurun_out/r05/j_soak.txt 2>&1; rc=$?; tail -3 gpurun_out/r05/j_soak.txt
[ $rc -ne 0 ] && exit $rc
echo "== bench lines, all configurations =="
bash tools/gpu_configs.sh && cp gpurun_out/configs.jsonl gpurun_out/r05/j_bench_all_configs.jsonl
