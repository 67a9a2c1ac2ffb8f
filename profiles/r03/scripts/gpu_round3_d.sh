#!/bin/bash
# round 3, GPU visit D: sparse waves scan together — parity, the end of the launch, A/B
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity =="
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_launch_paths.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_d.txt 2>&1; rc=$?; tail -8 gpurun_out/pytest_d.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
echo "== wave clocks =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_clocks.so timeout -k 10 120 python tools/gpu_wave_tail.py > gpurun_out/wave_tail_sparse.txt 2>&1 || { tail -5 gpurun_out/wave_tail_sparse.txt; exit 1; }
cat gpurun_out/wave_tail_sparse.txt
echo "== A/B: item queue without / with the cooperative scan of sparse waves (2 loads in flight) / with 4 loads in flight =="
timeout -k 10 500 python tools/gpu_ab.py synthetic-100k 1920 1080 64 2 librt_hip_nosparse.so librt_hip.so librt_hip_sparse4.so > gpurun_out/ab_sparse_100k.txt 2>&1 || { tail -5 gpurun_out/ab_sparse_100k.txt; exit 1; }
cat gpurun_out/ab_sparse_100k.txt
timeout -k 10 300 python tools/gpu_ab.py synthetic-10000 1920 1080 32 5 librt_hip_nosparse.so librt_hip.so librt_hip_sparse4.so > gpurun_out/ab_sparse_10k.txt 2>&1 || { tail -5 gpurun_out/ab_sparse_10k.txt; exit 1; }
cat gpurun_out/ab_sparse_10k.txt
echo "-- a low-resolution frame of the big scene (rt's preview-while-moving size): 240x135 at 16 spp"
timeout -k 10 300 python tools/gpu_ab.py synthetic-100k 240 135 16 3 librt_hip_nosparse.so librt_hip.so librt_hip_sparse4.so > gpurun_out/ab_sparse_lowres.txt 2>&1 || { tail -5 gpurun_out/ab_sparse_lowres.txt; exit 1; }
cat gpurun_out/ab_sparse_lowres.txt
