#!/bin/bash
# Round 5, visit AB: the closing tree (the LDS-resident kernel built per scan and camera form) — the whole GPU suite, rocprofv3 passes for every workload of
# profiles/pmc_counters.json (new source hash), one bench line per BASELINE configuration, the default bench.py line.
# (run as: bash profiles/r05/scripts/gpu_round5_w.sh from the repository root)
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r05/ab_pytest_gpu.txt 2>&1; rc=$?; tail -4 gpurun_out/r05/ab_pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh config5_streamed "--scene synthetic-100k --spp 64 --steps 2 --warmup 1" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh interactive_basic_plane_tilted "--scene basic_plane --tilt" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config3_dielectric "--scene dielectric" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config2_basic_64spp "--spp 64" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh config4_basic_4k "--width 3840 --height 2160" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh basic_plane_small "--scene basic_plane" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh basic_tilted_camera "--scene basic --tilt" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh dielectric_plane_small "--scene dielectric_plane" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh resident_64_spheres "--scene synthetic-64" || exit 1
echo "== bench lines: 9 to 32 spheres =="
: > gpurun_out/r05/ab_bench_small_resident.jsonl
for n in 9 12 16 24 32; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --no-interactive --no-kernel-only --steps 5 --warmup 2 --scene synthetic-$n --spp 64 | tail -1 >> gpurun_out/r05/ab_bench_small_resident.jsonl || exit 1
  tail -1 gpurun_out/r05/ab_bench_small_resident.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print($n, d['config'].get('kernel'), d['ms_per_step'], r['kernel_ms'], r['frac'])"
done
echo "== bench lines, all configurations =="
bash tools/gpu_configs.sh && cp gpurun_out/configs.jsonl gpurun_out/r05/ab_bench_all_configs.jsonl || exit 1
echo "== bench.py =="
timeout -k 10 300 python bench.py > gpurun_out/r05/ab_bench.jsonl 2> gpurun_out/r05/ab_bench.err; rc=$?; cut -c1-300 gpurun_out/r05/ab_bench.jsonl
exit $rc
