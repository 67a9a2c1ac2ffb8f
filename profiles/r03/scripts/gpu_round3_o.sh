#!/bin/bash
# Round 3, visit o: sub-chunk items in the big-scene kernels — parity, config 5 and its shares, the suite.
set -o pipefail
mkdir -p gpurun_out/o
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_half_chunks.py -x -q -m gpu > gpurun_out/o/pytest_half.txt 2>&1
echo "half-chunk tests: rc $?" | tee gpurun_out/o/status.txt
tail -12 gpurun_out/o/pytest_half.txt
grep -q " passed" gpurun_out/o/pytest_half.txt && ! grep -q "failed\|error" gpurun_out/o/pytest_half.txt || exit 1
timeout -k 10 300 python tools/gpu_config5_items.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/o/config5_items.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/o/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee -a gpurun_out/o/status.txt
tail -4 gpurun_out/o/pytest_gpu.txt
