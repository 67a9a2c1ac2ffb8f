#!/bin/bash
# round 3, GPU visit E: the whole suite on the final code, then the numbers for DESIGN.md / profiles/r03
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_check.sh || exit $?
cp gpurun_out/bench.txt gpurun_out/bench_headline_default.txt
echo "== every BASELINE configuration that fits one GPU =="
bash tools/gpu_configs.sh || exit $?
echo "== headline, RT_HIP_FLAG_FAST =="
timeout -k 10 300 python bench.py --fast --cpu-baseline-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_fast.txt || exit 1; cut -c1-250 gpurun_out/bench_fast.txt
echo "== four members on one device: gathered, direct =="
timeout -k 10 300 python bench.py --gpus 4 --same-device --cpu-baseline-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_4same.txt || exit 1; cut -c1-250 gpurun_out/bench_4same.txt
python -c "import json;d=json.load(open('gpurun_out/bench_4same.txt'));print(d['ms_per_step'], d['drop_in_breakdown'], d['per_rank'])"
timeout -k 10 300 python bench.py --gpus 4 --same-device --direct-frame --cpu-baseline-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_4same_direct.txt || exit 1
python -c "import json;d=json.load(open('gpurun_out/bench_4same_direct.txt'));print(d['ms_per_step'], d['drop_in_breakdown'], d['per_rank'])"
echo "== one rank under torchrun: both forms =="
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --cpu-baseline-seconds 0 2>/dev/null | tail -1 > gpurun_out/bench_torchrun1.txt || exit 1
python -c "import json;d=json.load(open('gpurun_out/bench_torchrun1.txt'));print(d['ms_per_step'], d['value_from'], d['paths'], d['drop_in_breakdown'], d['rccl'])"
echo "== shares of the headline frame =="
timeout -k 10 300 python tools/gpu_partition_times.py > gpurun_out/partition_times.txt 2>&1 || exit 1; cat gpurun_out/partition_times.txt
timeout -k 10 300 python tools/gpu_partition_times.py 64 > gpurun_out/partition_times_64spp.txt 2>&1 || exit 1; head -4 gpurun_out/partition_times_64spp.txt
echo "== host costs =="
timeout -k 10 200 python tools/gpu_host_cost.py > gpurun_out/host_cost.txt 2>&1 || exit 1; cat gpurun_out/host_cost.txt
