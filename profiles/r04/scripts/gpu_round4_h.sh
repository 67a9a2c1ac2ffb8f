#!/bin/bash
# Round 4, visit H (final kernels): the GPU suite ONCE; bench lines (plane scenes, tilted camera, both frame modes, every
# BASELINE configuration); rocprofv3 summaries — trace + PMC — of the headline and of the scenes the review asked for.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu (once) =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 500 -p no:cacheprovider > gpurun_out/r04/pytest_gpu.txt 2>&1; rc=$?; tail -12 gpurun_out/r04/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
echo "== bench lines =="
: > gpurun_out/r04/bench_planes.jsonl
for a in "--scene basic" "--scene basic_plane" "--scene basic_plane --resident" "--scene basic --resident" "--scene dielectric" "--scene dielectric_plane" "--scene dielectric_plane --resident" "--scene basic --tilt" "--scene basic --tilt --resident" "--scene basic_plane --tilt" "--scene synthetic-64 --spp 64" "--scene synthetic-64"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 $a >> gpurun_out/r04/bench_planes.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_planes.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$a', l['ms_per_step'], 'ms', l['roofline']['kernel'], l['roofline']['kernel_ms'], 'frac', l['roofline']['frac'])"
done
: > gpurun_out/r04/bench_all_configs.jsonl
for a in "" "--locked-frame" "--spp 64" "--scene dielectric" "--width 3840 --height 2160" "--width 256 --height 256 --spp 1" "--fast"; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 $a >> gpurun_out/r04/bench_all_configs.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_all_configs.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('[$a]', l['ms_per_step'], 'ms; kernel', l['roofline']['kernel_ms'], 'frac', l['roofline']['frac'], '; plug-in call', l.get('plug_in_call',{}).get('ms_per_step'), '; other mode', l.get('other_frame_mode',{}).get('ms_per_step'))"
done
echo "== profiles =="
bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
bash tools/gpu_profile_run.sh basic_plane_small "--scene basic_plane" || exit 1
bash tools/gpu_profile_run.sh basic_plane_resident "--scene basic_plane --resident" || exit 1
bash tools/gpu_profile_run.sh resident_64_spheres "--scene synthetic-64" || exit 1
bash tools/gpu_profile_run.sh basic_tilted_camera "--scene basic --tilt" || exit 1
echo "== several members on one device: default frame mode against the locked one =="
for a in "--gpus 4 --same-device" "--gpus 4 --same-device --locked-frame" "--gpus 4 --same-device --direct-frame" "--gpus 4 --same-device --direct-frame --locked-frame"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 $a >> gpurun_out/r04/bench_multi_member.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_multi_member.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('[$a]', l['ms_per_step'], 'ms', l['drop_in_breakdown'])"
done
exit 0
