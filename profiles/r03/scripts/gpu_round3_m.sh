#!/bin/bash
# Round 3, visit m: half-chunk items — parity, then the shares again, then the whole suite.
set -o pipefail
mkdir -p gpurun_out/m
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_half_chunks.py -x -q -m gpu > gpurun_out/m/pytest_half.txt 2>&1
echo "half-chunk tests: rc $?" | tee gpurun_out/m/status.txt
tail -15 gpurun_out/m/pytest_half.txt
grep -q " passed" gpurun_out/m/pytest_half.txt && ! grep -q "failed\|error" gpurun_out/m/pytest_half.txt || exit 1
timeout -k 10 300 python tools/gpu_far_share.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/m/far_share.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/m/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee -a gpurun_out/m/status.txt
tail -4 gpurun_out/m/pytest_gpu.txt
timeout -k 10 300 python tools/gpu_partition_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/m/partition_times.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-baseline-seconds 0 | cut -c1-240
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-baseline-seconds 0 --width 256 --height 256 --spp 16 | cut -c1-240
