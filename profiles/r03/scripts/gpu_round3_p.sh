#!/bin/bash
# Round 3, visit p: item size of the big-scene kernels chosen by scene size — parity, config 5 and friends, the suite.
set -o pipefail
mkdir -p gpurun_out/p
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_half_chunks.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/p/pytest_parity.txt 2>&1
echo "parity tests: rc $?" | tee gpurun_out/p/status.txt
tail -5 gpurun_out/p/pytest_parity.txt
grep -q " passed" gpurun_out/p/pytest_parity.txt && ! grep -q "failed\|error" gpurun_out/p/pytest_parity.txt || exit 1
timeout -k 10 300 python tools/gpu_config5_items.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/p/config5_items.txt
timeout -k 10 200 python bench.py --scene synthetic-100k --spp 64 --steps 2 --warmup 0 --settle-ms 0 --cpu-baseline-seconds 0 > gpurun_out/p/bench_c5.jsonl 2>&1
python -c "
import json; j=json.loads(open('gpurun_out/p/bench_c5.jsonl').read().strip().splitlines()[-1]); print('config 5 bench:', j['ms_per_step'], j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], (j.get('kernel_only') or {}).get('ms_per_step'))"
