#!/bin/bash
# Round 5, visit M: the stream counter taken from the window by the restarting lanes (one select; no copy per miss, no copies at the hit / miss join) on top of visit L (a lone plane's reciprocal unguarded, the counter advanced in place), against visits L and K; the whole GPU suite;
# region counters of the headline and of basic.toml with its plane; rocprofv3 passes; bench.py.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/m_pytest_gpu.txt 2>&1; rc=$?; tail -8 gpurun_out/r05/m_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
{
for cfg in "scenes/basic_plane.toml 1920 1080 256" "dielectric_plane 1920 1080 256" "basic 1920 1080 256" "synthetic-5 1920 1080 256"; do
  echo "== $cfg =="; timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so librt_hip_visit_l.so librt_hip_visit_k.so || exit 1
done
for cfg in "basic 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256"; do
  echo "== $cfg, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so librt_hip_visit_l.so librt_hip_visit_k.so || exit 1
done
} 2>&1 | tee gpurun_out/r05/m_ab.txt
echo "== region counters =="
for scene in basic scenes/basic_plane.toml; do
  RT_HIP_LIBRARY=rt_amd/lib/librt_hip_regions.so timeout -k 10 300 python tools/region_profile.py $scene 1920 1080 256 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r05/m_region_counters.txt
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh interactive_basic_plane_tilted "--scene basic_plane --tilt" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config3_dielectric "--scene dielectric" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config2_basic_64spp "--spp 64" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config4_basic_4k "--width 3840 --height 2160" || exit 1
echo "== bench.py =="
timeout -k 10 300 python bench.py > gpurun_out/r05/m_bench.jsonl 2> gpurun_out/r05/m_bench.err; rc=$?; cut -c1-300 gpurun_out/r05/m_bench.jsonl
exit $rc
