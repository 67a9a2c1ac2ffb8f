#!/bin/bash
# Round 5, visit G: the eye form of the general camera (any perspective matrix: per-pixel base, one reciprocal, no far point), the wave
# budget rule (6 waves from six spheres and seven primitives), the resident kernel scanning through the scalar cache (A/B), the
# region counters of the headline kernel, bench.py's line.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/g_pytest_gpu.txt 2>&1; rc=$?; tail -8 gpurun_out/r05/g_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
{
for cfg in "basic 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256" "dielectric 1920 1080 256" "dielectric_plane 1920 1080 256" "synthetic-8 1920 1080 256"; do
  echo "== $cfg =="; timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so || exit 1
done
echo "== basic 1920 1080 256, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py basic 1920 1080 256 15 librt_hip.so || exit 1
echo "== scenes/basic_plane.toml 1920 1080 256, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py scenes/basic_plane.toml 1920 1080 256 15 librt_hip.so || exit 1
echo "== dielectric 1920 1080 256, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py dielectric 1920 1080 256 15 librt_hip.so || exit 1
} 2>&1 | tee gpurun_out/r05/g_ab.txt
{
for cfg in "synthetic-12 1920 1080 64" "synthetic-24 1920 1080 64" "synthetic-64 1920 1080 64" "synthetic-64 1920 1080 256" "synthetic-200 1920 1080 64" "synthetic-700 1920 1080 64"; do
  echo "== $cfg =="; timeout -k 10 400 python tools/gpu_ab.py $cfg 8 librt_hip.so librt_hip_rscal.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/g_resident_scalar_ab.txt
echo "== region counters =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_regions.so timeout -k 10 300 python tools/region_profile.py basic 1920 1080 256 2>&1 | tee gpurun_out/r05/g_region_counters.txt
echo "== bench.py =="
timeout -k 10 300 python bench.py > gpurun_out/r05/g_bench.jsonl 2> gpurun_out/r05/g_bench.err; rc=$?; cut -c1-300 gpurun_out/r05/g_bench.jsonl
exit $rc
