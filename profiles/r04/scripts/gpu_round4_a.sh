#!/bin/bash
# Round 4, visit A (diagnosis only): what the HIP runtime does with pageable D2H destinations (round-3 library).
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp RT_HIP_LIBRARY=$PWD/rt_amd/lib/librt_hip_r3.so RT_HIP_DEBUG_FRAME=1
AMD_LOG_LEVEL=4 AMD_LOG_MASK=1792 timeout -k 10 300 python tools/gpu_pageable_diag.py > gpurun_out/r04/pageable_diag_stdout.txt 2> gpurun_out/r04/pageable_diag_amdlog.txt; rc=$?
echo "rc=$rc"; tail -30 gpurun_out/r04/pageable_diag_stdout.txt
grep -c "" gpurun_out/r04/pageable_diag_amdlog.txt
grep -E "Pinned resource|Staging resource|staging D2H|Unpinned" gpurun_out/r04/pageable_diag_amdlog.txt | sort | uniq -c | sort -rn | head -20
grep -E "Pinned|pinned|staging|Staging" gpurun_out/r04/pageable_diag_amdlog.txt | head -60 > gpurun_out/r04/pageable_diag_copy_paths.txt
# keep the merged log small
head -c 3000000 gpurun_out/r04/pageable_diag_amdlog.txt > gpurun_out/r04/pageable_diag_amdlog_head.txt; rm -f gpurun_out/r04/pageable_diag_amdlog.txt
exit 0
