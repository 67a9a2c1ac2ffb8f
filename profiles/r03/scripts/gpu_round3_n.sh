#!/bin/bash
# Round 3, visit n: 16-pixel tiles for big 256-spp frames in HBM too — the suite, the configurations, the headline.
set -o pipefail
mkdir -p gpurun_out/n
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/n/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee gpurun_out/n/status.txt
tail -4 gpurun_out/n/pytest_gpu.txt
grep -q " passed" gpurun_out/n/pytest_gpu.txt && ! grep -q "failed" gpurun_out/n/pytest_gpu.txt || exit 1
timeout -k 10 400 bash tools/gpu_configs.sh > gpurun_out/n/configs.log 2>&1; cp gpurun_out/configs.jsonl gpurun_out/n/configs.jsonl
timeout -k 10 300 python bench.py > gpurun_out/n/bench_default.jsonl 2> gpurun_out/n/bench_default.err
timeout -k 10 300 python tools/gpu_far_share.py 2>&1 | grep -v amdgpu.ids > gpurun_out/n/far_share.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/n/*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["config"]["workload"][:42], j["n_gpus"], j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], (j.get("kernel_only") or {}).get("ms_per_step"), (j.get("plug_in_call") or {}).get("ms_per_step"))
PY
cat gpurun_out/n/far_share.txt
