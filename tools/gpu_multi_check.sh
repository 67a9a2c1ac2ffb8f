#!/bin/bash
# multi-member context tests + the drop-in read-back A/B
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -q -x --timeout 300 > gpurun_out/pytest_multi.txt 2>&1; rc=$?; tail -15 gpurun_out/pytest_multi.txt
[ $rc -ne 0 ] && exit $rc
for mode in 1 0; do
  echo "== bench, store_to_host=$mode =="
