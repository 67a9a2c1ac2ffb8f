#!/bin/bash
# Round 3, visit l: after the tile change — the N > 1 forms again (torchrun with one rank at the headline size; ranks as
# processes on one device), host cost, partition times, the profile of the headline.
set -o pipefail
mkdir -p gpurun_out/l
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/l/bench_torchrun_1.jsonl 2> gpurun_out/l/bench_torchrun_1.err
echo "torchrun, one rank: rc $?" | tee gpurun_out/l/status.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/l/bench_single.jsonl 2>&1
for n in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29720 + n)) bench.py --gpus $n --steps 20 --warmup 3 --backend gloo --cpu-baseline-seconds 0 > gpurun_out/l/bench_gloo_$n.jsonl 2> gpurun_out/l/bench_gloo_$n.err
  echo "gloo rehearsal with $n processes: rc $?" | tee -a gpurun_out/l/status.txt
done
timeout -k 10 200 python bench.py --gpus 4 --same-device --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/l/bench_4_gathered.jsonl 2>&1
timeout -k 10 200 python bench.py --gpus 4 --same-device --direct-frame --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/l/bench_4_direct.jsonl 2>&1
timeout -k 10 200 python bench.py --fast --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/l/bench_fast.jsonl 2>&1
timeout -k 10 200 python tools/gpu_host_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/l/host_cost.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/l/bench_*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["n_gpus"], j["ms_per_step"], j["roofline"]["kernel_ms"], j.get("value_from"), {k: v.get("ms_per_step", v.get("status")) for k, v in (j.get("paths") or {}).items()}, j.get("drop_in_breakdown"))
PY
cat gpurun_out/l/host_cost.txt
