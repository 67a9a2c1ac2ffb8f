#!/bin/bash
# Round 5, visit Z: the dense build of the streamed kernel under the sm scatter table and with runs of 8 samples (the test cases added
# after visit Y), and config 5's tests once more.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "beyond_its_lds or config5 or handful_of_rays" > gpurun_out/r05/z_pytest_gpu.txt 2>&1; rc=$?; tail -5 gpurun_out/r05/z_pytest_gpu.txt
exit $rc
