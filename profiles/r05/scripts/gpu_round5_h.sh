#!/bin/bash
# Round 5, visit H: the kernels as they ship — the whole GPU suite, the A/B table of the round (kernel ms), resident against streamed
# around 1000 spheres, the headline's region counters, rocprofv3 passes of the headline, config 3 and the interactive workload, bench.py.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/h_pytest_gpu.txt 2>&1; rc=$?; tail -8 gpurun_out/r05/h_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
{
for cfg in "basic 1920 1080 256" "basic 1920 1080 64" "basic 3840 2160 256" "scenes/basic_plane.toml 1920 1080 256" "dielectric 1920 1080 256" "dielectric_plane 1920 1080 256" "synthetic-8 1920 1080 256" "synthetic-5 1920 1080 256" "basic 256 256 1"; do
  echo "== $cfg =="; timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so || exit 1
done
for cfg in "basic 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256" "dielectric 1920 1080 256"; do
  echo "== $cfg, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/h_final_ab.txt
{
for n in 700 1000; do
  echo "== synthetic-$n 1920 1080 64: the launch code's choice =="; timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 6 librt_hip.so || exit 1
  echo "== synthetic-$n 1920 1080 64: LDS-resident kernel forced =="; AB_FLAGS=2 timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 6 librt_hip.so || exit 1
  echo "== synthetic-$n 1920 1080 64: streamed kernel forced =="; AB_FLAGS=32 timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 6 librt_hip.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/h_resident_vs_streamed.txt
echo "== region counters =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_regions.so timeout -k 10 300 python tools/region_profile.py basic 1920 1080 256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/h_region_counters.txt
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config3_dielectric "--scene dielectric" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh interactive_basic_plane_tilted "--scene basic_plane --tilt" || exit 1
echo "== bench.py =="
timeout -k 10 300 python bench.py > gpurun_out/r05/h_bench.jsonl 2> gpurun_out/r05/h_bench.err; rc=$?; cut -c1-300 gpurun_out/r05/h_bench.jsonl
exit $rc
