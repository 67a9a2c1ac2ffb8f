#!/bin/bash
# Round 4, visit B: (1) the runtime's copy paths for pageable destinations, round-3 library, decimal log mask;
# (2) smoke + the whole GPU suite ONCE with the restructured library (module-owned frame + carrier, no caller memory handed
# to HIP); (3) the headline bench in both frame modes.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pageable diag (round-3 library) =="
RT_HIP_LIBRARY=$PWD/rt_amd/lib/librt_hip_r3.so AMD_LOG_LEVEL=4 AMD_LOG_MASK=1792 timeout -k 10 300 python tools/gpu_pageable_diag.py > gpurun_out/r04/pageable_diag_stdout.txt 2> gpurun_out/r04/pageable_diag_amdlog.txt; echo "rc=$?"
grep -E "Pinned resource|Staging resource|staging D2H|Unpinned|pinned" gpurun_out/r04/pageable_diag_amdlog.txt | sed -E 's/^[^]]*\] //; s/0x[0-9a-f]+/ADDR/g; s/[0-9]+ us/T us/' | sort | uniq -c | sort -rn | head -30 > gpurun_out/r04/pageable_diag_copy_paths.txt
cat gpurun_out/r04/pageable_diag_copy_paths.txt | head -12
grep -E "Pinned|pinned|Staging|staging" gpurun_out/r04/pageable_diag_amdlog.txt | head -80 > gpurun_out/r04/pageable_diag_copy_lines.txt
head -c 400000 gpurun_out/r04/pageable_diag_amdlog.txt > gpurun_out/r04/pageable_diag_amdlog_head.txt; rm -f gpurun_out/r04/pageable_diag_amdlog.txt
tail -12 gpurun_out/r04/pageable_diag_stdout.txt
echo "== smoke =="
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r04/smoke.txt 2>&1; rc=$?; tail -6 gpurun_out/r04/smoke.txt
[ $rc -ne 0 ] && exit $rc
echo "== pytest -m gpu (once) =="
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider > gpurun_out/r04/pytest_gpu.txt 2>&1; rc=$?; tail -30 gpurun_out/r04/pytest_gpu.txt
[ $rc -ge 124 ] && exit $rc
echo "== bench, default frame mode =="
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r04/bench_default.jsonl 2> gpurun_out/r04/bench_default.err; rc=$?; tail -c 3000 gpurun_out/r04/bench_default.jsonl; tail -3 gpurun_out/r04/bench_default.err
[ $rc -ge 124 ] && exit $rc
echo "== bench, locked frame mode =="
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --locked-frame --cpu-baseline-seconds 0 > gpurun_out/r04/bench_locked.jsonl 2> gpurun_out/r04/bench_locked.err; rc=$?; tail -c 1500 gpurun_out/r04/bench_locked.jsonl
exit 0
