#!/bin/bash
# the multi-member / rank context tests
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -q -x --timeout 300 > gpurun_out/pytest_multi.txt 2>&1; rc=$?; tail -15 gpurun_out/pytest_multi.txt
exit $rc
