#!/bin/bash
# Round 3, visit r: streamed kernel as the default for every big scene and from 705 primitives — the suite.
set -o pipefail
mkdir -p gpurun_out/r
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee gpurun_out/r/status.txt
tail -12 gpurun_out/r/pytest_gpu.txt
