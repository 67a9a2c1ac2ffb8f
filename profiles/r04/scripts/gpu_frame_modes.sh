#!/bin/bash
# The default frame mode (module-owned frame + carrier threads) against the zero-copy opt-in, per frame size and kernel length:
# the call as the plug-in makes it (stats == NULL) in both modes, from one bench.py process each.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
: > gpurun_out/r04/bench_frame_modes.jsonl
for a in "" "--spp 64" "--width 3840 --height 2160" "--scene dielectric" "--fast" "--width 800 --height 600 --spp 30" "--width 256 --height 256 --spp 1" "--spp 16"; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --steps 30 --warmup 3 $a >> gpurun_out/r04/bench_frame_modes.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_frame_modes.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('[$a] kernel %.4f | default: with stats %.4f, plug-in call %.4f | locked: plug-in call %.4f | default - locked = %+.4f ms' % (l['roofline']['kernel_ms'], l['ms_per_step'], l['plug_in_call']['ms_per_step'], l['other_frame_mode']['ms_per_step'], l['plug_in_call']['ms_per_step'] - l['other_frame_mode']['ms_per_step']))" | tee -a gpurun_out/r04/frame_modes_summary.txt
done
