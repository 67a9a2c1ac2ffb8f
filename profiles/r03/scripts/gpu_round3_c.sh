#!/bin/bash
# round 3, GPU visit C: the item queue of the big-scene kernels — parity first, then how the launch ends and what it buys
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity: every test that runs a big-scene kernel =="
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_launch_paths.py tests/test_gpu_fast.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_c.txt 2>&1; rc=$?; tail -8 gpurun_out/pytest_c.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
echo "== wave clocks, item queue =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_clocks.so timeout -k 10 120 python tools/gpu_wave_tail.py > gpurun_out/wave_tail_items.txt 2>&1 || { tail -5 gpurun_out/wave_tail_items.txt; exit 1; }
cat gpurun_out/wave_tail_items.txt
echo "== A/B: 64-item tiles (round 3, before) vs item queue =="
timeout -k 10 400 python tools/gpu_ab.py synthetic-100k 1920 1080 64 2 librt_hip_r3tiles64.so librt_hip.so > gpurun_out/ab_items_100k.txt 2>&1 || { tail -5 gpurun_out/ab_items_100k.txt; exit 1; }
cat gpurun_out/ab_items_100k.txt
timeout -k 10 300 python tools/gpu_ab.py synthetic-10000 1920 1080 32 5 librt_hip_r3tiles64.so librt_hip.so > gpurun_out/ab_items_10k.txt 2>&1 || { tail -5 gpurun_out/ab_items_10k.txt; exit 1; }
cat gpurun_out/ab_items_10k.txt
timeout -k 10 300 python tools/gpu_ab.py synthetic-2000 1920 1080 64 5 librt_hip_r3tiles64.so librt_hip.so > gpurun_out/ab_items_2k.txt 2>&1 || { tail -5 gpurun_out/ab_items_2k.txt; exit 1; }
cat gpurun_out/ab_items_2k.txt
AB_FLAGS=1 timeout -k 10 300 python tools/gpu_ab.py synthetic-10000 1920 1080 8 5 librt_hip_r3tiles64.so librt_hip.so > gpurun_out/ab_items_10k_tiled8.txt 2>&1 || { tail -5 gpurun_out/ab_items_10k_tiled8.txt; exit 1; }
cat gpurun_out/ab_items_10k_tiled8.txt
