#!/bin/bash
# round 3, GPU visit A: kernel facilities probe, the whole GPU suite, bench, streamed-kernel prefetch A/B, clocks of config 5
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
uname -r > gpurun_out/vma_probe.txt; ./tools/vma_probe >> gpurun_out/vma_probe.txt 2>&1; cat gpurun_out/vma_probe.txt
bash tools/gpu_check.sh || exit $?
echo "== streamed kernel: next group's load issued before this group's probes (default) vs not =="
timeout -k 10 400 python tools/gpu_ab.py synthetic-100k 1920 1080 64 2 librt_hip_nopf.so librt_hip.so > gpurun_out/ab_prefetch_100k.txt 2>&1 || { tail -5 gpurun_out/ab_prefetch_100k.txt; exit 1; }
cat gpurun_out/ab_prefetch_100k.txt
timeout -k 10 300 python tools/gpu_ab.py synthetic-10000 1920 1080 32 5 librt_hip_nopf.so librt_hip.so > gpurun_out/ab_prefetch_10k.txt 2>&1 || { tail -5 gpurun_out/ab_prefetch_10k.txt; exit 1; }
cat gpurun_out/ab_prefetch_10k.txt
echo "== clocks of THIS GPU around 4 launches of config 5 =="
mkdir -p gpurun_out/c5
timeout -k 10 200 python3 tools/gpu_clock_sampler.py gpurun_out/c5/clocks_back_to_back.csv 100 -- python3 tools/gpu_config5_launches.py 4 0 > gpurun_out/c5/launches_back_to_back.txt 2>&1 || { tail -20 gpurun_out/c5/launches_back_to_back.txt; exit 1; }
grep launch gpurun_out/c5/launches_back_to_back.txt
head -3 gpurun_out/c5/clocks_back_to_back.csv
