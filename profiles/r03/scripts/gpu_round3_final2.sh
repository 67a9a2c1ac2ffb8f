#!/bin/bash
# Round 3, last visit: the whole GPU suite, the headline's profile, traffic counters of the other configurations, every
# bench line that DESIGN.md quotes.
set -o pipefail
mkdir -p gpurun_out/z; rm -rf gpurun_out/z/*
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/z/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee gpurun_out/z/status.txt
tail -4 gpurun_out/z/pytest_gpu.txt
grep -q " passed" gpurun_out/z/pytest_gpu.txt && ! grep -q "failed" gpurun_out/z/pytest_gpu.txt || exit 1
rm -rf gpurun_out/prof; bash tools/gpu_profile.sh > gpurun_out/z/profile.log 2>&1; echo "profile: rc $?" | tee -a gpurun_out/z/status.txt
bash tools/gpu_traffic_configs.sh > gpurun_out/z/traffic.log 2>&1; echo "traffic passes: rc $?" | tee -a gpurun_out/z/status.txt
timeout -k 10 400 bash tools/gpu_configs.sh > gpurun_out/z/configs.log 2>&1; cp gpurun_out/configs.jsonl gpurun_out/z/configs.jsonl
timeout -k 10 300 python bench.py > gpurun_out/z/bench_default.jsonl 2> gpurun_out/z/bench_default.err; echo "default bench: rc $?" | tee -a gpurun_out/z/status.txt
timeout -k 10 200 python bench.py --fast --cpu-baseline-seconds 0 > gpurun_out/z/bench_fast.jsonl 2>&1
timeout -k 10 200 python bench.py --gpus 4 --same-device --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/z/bench_4_gathered.jsonl 2>&1
timeout -k 10 200 python bench.py --gpus 4 --same-device --direct-frame --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/z/bench_4_direct.jsonl 2>&1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 1 --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/z/bench_torchrun_1.jsonl 2> gpurun_out/z/bench_torchrun_1.err
for n in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29820 + n)) bench.py --gpus $n --steps 20 --warmup 3 --backend gloo --cpu-baseline-seconds 0 > gpurun_out/z/bench_gloo_$n.jsonl 2> gpurun_out/z/bench_gloo_$n.err
done
timeout -k 10 300 python tools/gpu_partition_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/z/partition_times.txt
timeout -k 10 200 python tools/gpu_host_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/z/host_cost.txt
timeout -k 10 300 python tools/gpu_far_share.py 2>&1 | grep -v amdgpu.ids > gpurun_out/z/far_share.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/z/*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["config"]["workload"][:42], j["n_gpus"], j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], (j.get("kernel_only") or {}).get("ms_per_step"), (j.get("plug_in_call") or {}).get("ms_per_step"), j.get("value_from"), {k: v.get("ms_per_step", v.get("status")) for k, v in (j.get("paths") or {}).items()})
PY
cat gpurun_out/z/partition_times.txt | head -12
