#!/bin/bash
# Round 4, visit S: the assemble kernel with one 16-byte store per lane (experiment build) against 4-byte stores: several members
# on ONE device, where assemble_ms is the kernel's own time for (n-1)/n of the frame over PCIe.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
for round in 1 2; do
for lib in librt_hip.so librt_hip_asm4.so; do
  for n in 4 8; do
    for mode in "" "--locked-frame"; do
      RT_HIP_LIBRARY=$PWD/rt_amd/lib/$lib timeout -k 10 300 python bench.py --gpus $n --same-device --steps 20 --warmup 3 --cpu-baseline-seconds 0 $mode 2>/tmp/err.txt | python -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); b=l['drop_in_breakdown']
print('$lib n=$n $mode'.ljust(44), 'wall', l['ms_per_step'], 'kernel', b['kernel_ms'], 'gather', b['gather_ms'], 'assemble', b['assemble_ms'])" || { tail -3 /tmp/err.txt; exit 1; }
    done
  done
done
done 2>&1 | tee gpurun_out/r04/assemble_x4_ab.txt
