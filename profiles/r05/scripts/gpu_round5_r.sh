#!/bin/bash
# Round 5, visit R: the scenes between the two tuned ends, final kernels: one bench.py line per sphere count from
# 9 to 2000 at 1080p x 64 spp — the LDS-resident kernel (LDS scan below 40 spheres, scalar-load scan from 40 on) up to 1024
# primitives, the streamed kernel beyond — with the fraction of the FP32 vector peak each reaches; and a longer soak of random scenes.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
out=gpurun_out/r05/r_midsize.jsonl; : > $out
for n in 9 12 16 24 32 40 64 128 200 400 700 1000 1024 1100 2000; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --no-interactive --no-kernel-only --steps 5 --warmup 2 --scene synthetic-$n --spp 64 | tail -1 >> $out || exit 1
  tail -1 $out | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$n', d['config'].get('kernel'), d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('mean_segments_per_sample'))"
done
echo "== soak: 10 000 random scenes x 11 modes =="
RT_HIP_RANDOM_CASES=10000 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k random_scenes > gpurun_out/r05/r_soak_random_scenes.txt 2>&1; rc=$?; tail -2 gpurun_out/r05/r_soak_random_scenes.txt
exit $rc
