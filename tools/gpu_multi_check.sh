#!/bin/bash
# multi-member context tests + the drop-in read-back A/B
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -q -x --timeout 300 > gpurun_out/pytest_multi.txt 2>&1; rc=$?; tail -15 gpurun_out/pytest_multi.txt
[ $rc -ne 0 ] && exit $rc
for mode in 1 0; do
  echo "== bench, store_to_host=$mode =="
  RT_HIP_EXPERIMENT_STORE_TO_HOST=$mode timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/bench_store$mode.txt 2>&1 || { tail -5 gpurun_out/bench_store$mode.txt; exit 1; }
  python3 -c "
import json,sys
l=[x for x in open('gpurun_out/bench_store$mode.txt') if x.startswith('{')][-1]
d=json.loads(l); print('kernel_ms',d['roofline']['kernel_ms'],'ms_per_step',d['ms_per_step'],'drop_in',d['drop_in_render'])"
done
