#!/bin/bash
# Round 4, visit D2: A/B series.  (1) lane modes as scalar masks vs the round-3 mode register (the one
# structural experiment); (2) one 16-byte store/load per parked value vs the 8 + 4 byte pair, and the memory-model
# ordering of the arrival vs the ISA-level one; (3) config 5's counters (WRITE_SIZE).
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
ab() { out=$1; shift; echo "== $* ==" | tee -a gpurun_out/r04/$out; timeout -k 10 600 python tools/gpu_ab.py "$@" 2>&1 | tee -a gpurun_out/r04/$out; }
: > gpurun_out/r04/ab_mode_masks.txt; : > gpurun_out/r04/arrival_ordering_ab.txt; : > gpurun_out/r04/ab_publish_16_bytes.txt
ab ab_mode_masks.txt basic 1920 1080 256 30 librt_hip_modereg.so librt_hip.so
ab ab_mode_masks.txt dielectric 1920 1080 256 30 librt_hip_modereg.so librt_hip.so
ab ab_mode_masks.txt basic 1920 1080 64 40 librt_hip_modereg.so librt_hip.so
ab ab_mode_masks.txt basic_plane 1920 1080 256 30 librt_hip_modereg.so librt_hip.so
ab ab_mode_masks.txt synthetic-64 1920 1080 64 20 librt_hip_modereg.so librt_hip.so
ab arrival_ordering_ab.txt synthetic-10000 1920 1080 32 5 librt_hip_split.so librt_hip_model.so librt_hip.so
ab arrival_ordering_ab.txt synthetic-2000 1920 1080 64 5 librt_hip_split.so librt_hip_model.so librt_hip.so
ab ab_publish_16_bytes.txt synthetic-100k 1920 1080 64 2 librt_hip_split.so librt_hip.so
echo "== config 5 counters =="
bash tools/gpu_profile_r4.sh config5_streamed "--scene synthetic-100k --spp 64 --steps 2 --warmup 1" || exit 1
exit 0
