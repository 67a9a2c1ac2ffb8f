#!/bin/bash
# What in the default frame mode costs what, each setting three times (fresh processes): helper threads pinned to the GPU's
# host node or not, x streamed zeros, on a long (4K) and the headline frame.  default - locked in ms per call (stats == NULL).
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
: > gpurun_out/r04/carrier_matrix.txt
for rep in 1 2 3 4; do for pin in 1 0; do for zeros in 1; do
  for a in "--width 3840 --height 2160" ""; do
    RT_HIP_CARRIER_PIN=$pin RT_HIP_CARRIER_STREAM_ZEROS=$zeros timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --steps 20 --warmup 2 --settle-ms 50 $a 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('pin=$pin stream_zeros=$zeros rep $rep [$a] kernel %.4f default %.4f locked %.4f diff %+.4f' % (l['roofline']['kernel_ms'], l['plug_in_call']['ms_per_step'], l['other_frame_mode']['ms_per_step'], l['plug_in_call']['ms_per_step'] - l['other_frame_mode']['ms_per_step']))" | tee -a gpurun_out/r04/carrier_matrix.txt
  done
done; done; done
