#!/bin/bash
# Round 3, visit j: stripes on their owners' NUMA nodes — the frame group tests, then what far memory costs a share.
set -o pipefail
mkdir -p gpurun_out/j
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_frame_group.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/j/pytest.txt 2>&1
echo "frame group + multi tests: rc $?" | tee gpurun_out/j/status.txt
tail -8 gpurun_out/j/pytest.txt
