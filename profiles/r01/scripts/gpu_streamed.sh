#!/bin/bash
# A/B of the two kernels for scenes that do not fit LDS: LDS-tiled vs scalar-streamed, synthetic spheres at 1080p
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 300 -k "frame_is_bit_exact or random_scenes" > gpurun_out/pytest_streamed.txt 2>&1; rc=$?; tail -5 gpurun_out/pytest_streamed.txt
[ $rc -ne 0 ] && exit $rc
for scene in synthetic-100k synthetic-10000 synthetic-2000; do
  for mode in --tiled --streamed; do
    spp=8; [ $scene = synthetic-2000 ] && spp=64; [ $scene = synthetic-10000 ] && spp=32
    echo "== $scene $mode spp $spp" | tee -a gpurun_out/streamed_ab.txt
    timeout -k 10 300 python bench.py --scene $scene --spp $spp --steps 2 --warmup 1 $mode --cpu-baseline-seconds 0 2>/dev/null | tee -a gpurun_out/streamed_ab.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['config']['kernel'])" || exit 1
  done
done
