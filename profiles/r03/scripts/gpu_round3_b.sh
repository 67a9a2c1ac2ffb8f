#!/bin/bash
# round 3, GPU visit B: how the persistent launch ends (wave clocks), tile size A/B, host costs, the changed tests
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== wave clocks, 128-item tiles =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_clocks.so timeout -k 10 120 python tools/gpu_wave_tail.py > gpurun_out/wave_tail_128.txt 2>&1 || { tail -5 gpurun_out/wave_tail_128.txt; exit 1; }
cat gpurun_out/wave_tail_128.txt
echo "== wave clocks, 64-item tiles =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_clocks64.so timeout -k 10 120 python tools/gpu_wave_tail.py > gpurun_out/wave_tail_64.txt 2>&1 || { tail -5 gpurun_out/wave_tail_64.txt; exit 1; }
cat gpurun_out/wave_tail_64.txt
echo "== tile size A/B =="
timeout -k 10 400 python tools/gpu_ab.py synthetic-100k 1920 1080 64 2 librt_hip.so librt_hip_tile64.so > gpurun_out/ab_tile64_100k.txt 2>&1 || { tail -5 gpurun_out/ab_tile64_100k.txt; exit 1; }
cat gpurun_out/ab_tile64_100k.txt
timeout -k 10 300 python tools/gpu_ab.py synthetic-10000 1920 1080 32 5 librt_hip.so librt_hip_tile64.so > gpurun_out/ab_tile64_10k.txt 2>&1 || { tail -5 gpurun_out/ab_tile64_10k.txt; exit 1; }
cat gpurun_out/ab_tile64_10k.txt
echo "== host costs =="
timeout -k 10 200 python tools/gpu_host_cost.py > gpurun_out/host_cost.txt 2>&1 || { tail -5 gpurun_out/host_cost.txt; exit 1; }
cat gpurun_out/host_cost.txt
echo "== changed tests =="
timeout -k 10 600 python -m pytest tests/test_bench_contract.py tests/test_gpu_fast.py -m gpu -q -x --timeout 300 -s > gpurun_out/pytest_b.txt 2>&1; rc=$?; tail -25 gpurun_out/pytest_b.txt | cut -c1-400
exit $rc
