#!/bin/bash
# Round 5, visit I: the kernels after the move / guard cuts (hit point straight into the ray, mode and bounce count in one word,
# the pinhole vector carried as near - eye, plain eye-form frames without guard and flip) — suite, A/B table, region counters,
# rocprofv3 passes, bench.py.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/i_pytest_gpu.txt 2>&1; rc=$?; tail -8 gpurun_out/r05/i_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
{
for cfg in "basic 1920 1080 256" "basic 1920 1080 64" "basic 3840 2160 256" "scenes/basic_plane.toml 1920 1080 256" "dielectric 1920 1080 256" "dielectric_plane 1920 1080 256" "synthetic-8 1920 1080 256" "synthetic-5 1920 1080 256" "synthetic-64 1920 1080 256" "synthetic-12 1920 1080 64" "basic 256 256 1"; do
  echo "== $cfg =="; timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so || exit 1
done
for cfg in "basic 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256"; do
  echo "== $cfg, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/i_final_ab.txt
echo "== region counters =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_regions.so timeout -k 10 300 python tools/region_profile.py basic 1920 1080 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/i_region_counters.txt
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config3_dielectric "--scene dielectric" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh interactive_basic_plane_tilted "--scene basic_plane --tilt" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config2_basic_64spp "--spp 64" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config4_basic_4k "--width 3840 --height 2160" || exit 1
echo "== bench.py =="
timeout -k 10 300 python bench.py > gpurun_out/r05/i_bench.jsonl 2> gpurun_out/r05/i_bench.err; rc=$?; cut -c1-300 gpurun_out/r05/i_bench.jsonl
exit $rc
