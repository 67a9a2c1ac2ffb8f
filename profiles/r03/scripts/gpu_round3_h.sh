#!/bin/bash
# Round 3, visit h: the frame group (rt_hip_join_frame_group) — its tests, the bench contract tests that touch the N > 1
# flow, and the rehearsal of the three forms with four rank processes on the one GPU.
set -o pipefail
mkdir -p gpurun_out/h
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_frame_group.py -x -q -m gpu > gpurun_out/h/pytest_frame_group.txt 2>&1
echo "frame group tests: rc $?" | tee gpurun_out/h/status.txt
tail -5 gpurun_out/h/pytest_frame_group.txt
grep -q "passed" gpurun_out/h/pytest_frame_group.txt && ! grep -q "failed" gpurun_out/h/pytest_frame_group.txt || exit 1
timeout -k 10 500 python -m pytest tests/test_bench_contract.py -x -q -m gpu > gpurun_out/h/pytest_bench_contract.txt 2>&1
echo "bench contract tests: rc $?" | tee -a gpurun_out/h/status.txt
tail -5 gpurun_out/h/pytest_bench_contract.txt
for n in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 20 --warmup 3 --backend gloo --cpu-baseline-seconds 0 > gpurun_out/h/bench_gloo_$n.jsonl 2> gpurun_out/h/bench_gloo_$n.err
  echo "gloo rehearsal with $n processes: rc $?" | tee -a gpurun_out/h/status.txt
done
timeout -k 10 200 python bench.py --gpus 4 --same-device --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/h/bench_same_device_gathered.jsonl 2>&1
timeout -k 10 200 python bench.py --gpus 4 --same-device --direct-frame --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/h/bench_same_device_direct.jsonl 2>&1
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/h/bench_single.jsonl 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/h/bench_*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["n_gpus"], j["ms_per_step"], j.get("value_from"), {k: v.get("ms_per_step", v.get("status")) for k, v in (j.get("paths") or {}).items()}, j.get("drop_in_breakdown"))
PY
