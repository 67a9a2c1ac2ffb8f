#!/bin/bash
# round 3, GPU visit G: contract v3's affine primary rays — parity, then A/B against the build before them
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_preview.py tests/test_headless.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_g.txt 2>&1; rc=$?; tail -6 gpurun_out/pytest_g.txt | cut -c1-300
[ $rc -ne 0 ] && exit $rc
for cfg in "basic 1920 1080 256" "dielectric 1920 1080 256" "basic 1920 1080 64"; do
  echo "== $cfg: before / after =="
  timeout -k 10 300 python tools/gpu_ab.py $cfg 30 librt_hip_precam.so librt_hip.so || exit 1
done 2>&1 | tee gpurun_out/ab_affine_rays.txt
echo "== config 5 and 10k spheres: before / after =="
timeout -k 10 400 python tools/gpu_ab.py synthetic-100k 1920 1080 64 2 librt_hip_precam.so librt_hip.so 2>&1 | tee -a gpurun_out/ab_affine_rays.txt
timeout -k 10 300 python tools/gpu_ab.py synthetic-10000 1920 1080 32 5 librt_hip_precam.so librt_hip.so 2>&1 | tee -a gpurun_out/ab_affine_rays.txt
timeout -k 10 300 python tools/gpu_ab.py synthetic-1000 1920 1080 64 10 librt_hip_precam.so librt_hip.so 2>&1 | tee -a gpurun_out/ab_affine_rays.txt
