#!/bin/bash
# Round 3, visit k: tiles cut for the destination (wide rows into host memory) — parity first, then the shares again.
set -o pipefail
mkdir -p gpurun_out/k
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/k/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee gpurun_out/k/status.txt
tail -4 gpurun_out/k/pytest_gpu.txt
grep -q " passed" gpurun_out/k/pytest_gpu.txt && ! grep -q "failed" gpurun_out/k/pytest_gpu.txt || exit 1
timeout -k 10 300 python tools/gpu_far_share.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k/far_share.txt
timeout -k 10 300 bash tools/gpu_configs.sh > gpurun_out/k/configs.log 2>&1
cp gpurun_out/configs.jsonl gpurun_out/k/configs.jsonl 2>/dev/null
timeout -k 10 200 python bench.py --steps 20 --warmup 3 > gpurun_out/k/bench_single.jsonl 2>gpurun_out/k/bench_single.err
timeout -k 10 200 python bench.py --gpus 8 --same-device --direct-frame --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/k/bench_8_direct.jsonl 2>&1
timeout -k 10 200 python bench.py --gpus 8 --same-device --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/k/bench_8_gathered.jsonl 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/k/*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["config"]["workload"][:40], j["n_gpus"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], (j.get("kernel_only") or {}).get("ms_per_step"), (j.get("plug_in_call") or {}).get("ms_per_step"))
PY
