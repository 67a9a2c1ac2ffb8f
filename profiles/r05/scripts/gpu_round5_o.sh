#!/bin/bash
# Round 5, visit O: rocprofv3 passes of the final kernels (sources 984e72e903153e41) for the workloads whose counter entries in
# profiles/pmc_counters.json still date from round 4 — the plane scenes through the axis-aligned camera, the tilted camera without a
# plane, 64 spheres in the LDS-resident kernel, config 5 — and eight consecutive default `python bench.py` runs (the carrier's
# figures per run: bands_early, helpers, host CPU, cgroup throttling).
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh basic_plane_small "--scene basic_plane" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh basic_tilted_camera "--scene basic --tilt" || exit 1
ROUND=r05 bash tools/gpu_profile_run.sh dielectric_plane_small "--scene dielectric_plane" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh resident_64_spheres "--scene synthetic-64" || exit 1
ROUND=r05 MIX=1 bash tools/gpu_profile_run.sh config5_streamed "--scene synthetic-100k --spp 64 --steps 2 --warmup 1" || exit 1
echo "== eight default runs =="
: > gpurun_out/r05/o_default_mode_repeats.jsonl
for i in 1 2 3 4 5 6 7 8; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --no-interactive | tail -1 >> gpurun_out/r05/o_default_mode_repeats.jsonl || exit 1
  tail -1 gpurun_out/r05/o_default_mode_repeats.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['drop_in_breakdown']; print(d['ms_per_step'], b['kernel_ms'], b['after_kernel_ms'], b['bands_early'], b['bands_early_min'], b['helpers'], b['host_cpu_ms_per_step'], b['cgroup_throttled'], d['frame_matches_oracle'])"
done
