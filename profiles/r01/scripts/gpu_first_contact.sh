#!/bin/bash
# First GPU contact: microbench, smoke, parity tests, a short bench, a rocprofv3 kernel trace.
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== host =="; nproc; grep -m1 "model name" /proc/cpuinfo
echo "== microbench ==" 
timeout -k 10 120 ./tools/microbench > gpurun_out/microbench.txt 2>&1 || { echo "microbench failed"; cat gpurun_out/microbench.txt; exit 1; }
cat gpurun_out/microbench.txt
echo "== smoke =="
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; rc=$?; tail -5 gpurun_out/smoke.txt
[ $rc -ge 124 ] && exit $rc
echo "== pytest -m gpu =="
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 300 > gpurun_out/pytest_gpu.txt 2>&1; rc=$?; tail -25 gpurun_out/pytest_gpu.txt
[ $rc -ge 124 ] && exit $rc
echo "== bench =="
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.txt 2>&1; rc=$?; tail -3 gpurun_out/bench.txt
[ $rc -ge 124 ] && exit $rc
exit 0
