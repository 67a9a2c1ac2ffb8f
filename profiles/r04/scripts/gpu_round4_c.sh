#!/bin/bash
# Round 4, visit C: runtime copy-path log (round-3 library); the whole GPU suite ONCE (planes in the scalar-register kernel,
# module-owned frame); bench lines of the reference's scenes with their ground plane, through the scalar-register kernel and
# through the LDS-resident one; rocprofv3 summaries (trace + PMC) of the headline, the plane scenes and a 64-sphere scene.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu (once) =="
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider > gpurun_out/r04/pytest_gpu.txt 2>&1; rc=$?; tail -30 gpurun_out/r04/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
echo "== bench lines: plane scenes =="
: > gpurun_out/r04/bench_planes.jsonl
for a in "--scene basic" "--scene basic_plane" "--scene basic_plane --resident" "--scene dielectric" "--scene dielectric_plane" "--scene dielectric_plane --resident" "--scene synthetic-64 --spp 64" "--scene synthetic-64"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 $a >> gpurun_out/r04/bench_planes.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_planes.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$a', l['ms_per_step'], 'ms', l['roofline']['kernel'], l['roofline']['kernel_ms'], 'frac', l['roofline']['frac'])"
done
echo "== profiles =="
bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
bash tools/gpu_profile_run.sh basic_plane_small "--scene basic_plane" || exit 1
bash tools/gpu_profile_run.sh basic_plane_resident "--scene basic_plane --resident" || exit 1
bash tools/gpu_profile_run.sh resident_64_spheres "--scene synthetic-64" || exit 1
exit 0
