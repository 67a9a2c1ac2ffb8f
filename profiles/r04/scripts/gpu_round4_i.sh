#!/bin/bash
# Round 4, visit I (final host side; kernels as profiled in visit H): the GPU suite ONCE; the default frame mode against the
# zero-copy opt-in per frame size; bench lines of every BASELINE configuration and of several members on one device.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu (once) =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 500 -p no:cacheprovider > gpurun_out/r04/pytest_gpu.txt 2>&1; rc=$?; tail -12 gpurun_out/r04/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/r04/frame_modes_summary.txt
bash tools/gpu_frame_modes.sh || exit 1
: > gpurun_out/r04/bench_all_configs.jsonl
for a in "" "--locked-frame" "--spp 64" "--scene dielectric" "--width 3840 --height 2160" "--width 256 --height 256 --spp 1" "--fast" "--scene synthetic-100k --spp 64 --steps 2 --warmup 1"; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 $a >> gpurun_out/r04/bench_all_configs.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_all_configs.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('[$a]', l['ms_per_step'], 'ms; kernel', l['roofline']['kernel_ms'], 'frac', l['roofline']['frac'], 'issue', (l['roofline'].get('issue') or {}).get('frac_of_issue_ceiling'), '; plug-in call', l.get('plug_in_call',{}).get('ms_per_step'), '; other mode', l.get('other_frame_mode',{}).get('ms_per_step'))"
done
: > gpurun_out/r04/bench_multi_member.jsonl
for a in "--gpus 4 --same-device" "--gpus 4 --same-device --locked-frame" "--gpus 4 --same-device --direct-frame" "--gpus 4 --same-device --direct-frame --locked-frame"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 $a >> gpurun_out/r04/bench_multi_member.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_multi_member.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('[$a]', l['ms_per_step'], 'ms', l['drop_in_breakdown'])"
done
echo "== the default bench line (what the driver runs) =="
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_headline.jsonl 2>/tmp/bench.err; tail -c 1800 gpurun_out/r04/bench_headline.jsonl
exit 0
