#!/bin/bash
# Round 4, visit G: is the rolling kernels' 8 % against round 3's figures the code or the box?  The round-3 build
# (rt_amd/lib/librt_hip_r3.so, kept from the start of the round) against today's, same box, interleaved.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
ab() { out=$1; shift; echo "== $* ==" | tee -a gpurun_out/r04/$out; timeout -k 10 600 python tools/gpu_ab.py "$@" 2>&1 | tee -a gpurun_out/r04/$out; }
: > gpurun_out/r04/round3_vs_round4_ab.txt
ab round3_vs_round4_ab.txt synthetic-10000 1920 1080 32 5 librt_hip_r3.so librt_hip.so
ab round3_vs_round4_ab.txt synthetic-2000 1920 1080 64 5 librt_hip_r3.so librt_hip.so
ab round3_vs_round4_ab.txt basic 1920 1080 256 30 librt_hip_r3.so librt_hip.so
ab round3_vs_round4_ab.txt dielectric 1920 1080 256 30 librt_hip_r3.so librt_hip.so
ab round3_vs_round4_ab.txt synthetic-100k 1920 1080 64 1 librt_hip_r3.so librt_hip.so
echo "== how far ahead of the drain is the carrier? =="
RT_HIP_DEBUG_FRAME=1 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --cpu-baseline-seconds 0 --no-kernel-only 2>&1 >/dev/null | grep "frame delivered" | tail -5 | tee gpurun_out/r04/carrier_early_bands.txt
exit 0
