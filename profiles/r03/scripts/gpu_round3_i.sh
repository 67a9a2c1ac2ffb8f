#!/bin/bash
# Round 3, visit i: the whole GPU suite with ABI 5, then what the frame group costs with a world of one (torchrun, one
# rank: all three forms at the headline size) next to the single-GPU call.
set -o pipefail
mkdir -p gpurun_out/i
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/i/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee gpurun_out/i/status.txt
tail -4 gpurun_out/i/pytest_gpu.txt
grep -q " passed" gpurun_out/i/pytest_gpu.txt && ! grep -q "failed" gpurun_out/i/pytest_gpu.txt || exit 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/i/bench_torchrun_1.jsonl 2> gpurun_out/i/bench_torchrun_1.err
echo "torchrun, one rank: rc $?" | tee -a gpurun_out/i/status.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/i/bench_single.jsonl 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/i/bench_*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["n_gpus"], j["ms_per_step"], j.get("value_from"), {k: v.get("ms_per_step", v.get("status")) for k, v in (j.get("paths") or {}).items()}, j.get("drop_in_breakdown"))
PY
