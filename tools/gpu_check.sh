#!/bin/bash
# parity tests + short bench (+ optional rocprof stats) on the GPU box
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke =="
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; rc=$?; tail -4 gpurun_out/smoke.txt
[ $rc -ne 0 ] && exit $rc
echo "== pytest -m gpu =="
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/pytest_gpu.txt 2>&1; rc=$?; tail -15 gpurun_out/pytest_gpu.txt
[ $rc -ge 124 ] && exit $rc
echo "== bench =="
timeout -k 10 300 python bench.py --steps 10 --warmup 3 $BENCH_ARGS > gpurun_out/bench.txt 2>&1; rc=$?; tail -2 gpurun_out/bench.txt
exit $rc
