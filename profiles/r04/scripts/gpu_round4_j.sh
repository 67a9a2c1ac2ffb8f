#!/bin/bash
# Round 4, visit J: config 5's counters with the final kernels; the line bench.py prints under torchrun (N = 1: everything
# but the second GPU), and with four rank processes on the one GPU over gloo.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== torchrun, 1 rank =="
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r04/bench_torchrun1.jsonl 2> gpurun_out/r04/bench_torchrun1.err; echo "rc=$?"; python -c "
import json
l=json.loads([x for x in open('gpurun_out/r04/bench_torchrun1.jsonl') if x.startswith('{')][-1])
print('value_from', l['value_from'], 'ms_per_step', l['ms_per_step'], 'rccl', l['rccl']); print({k:(v.get('status'), v.get('ms_per_step')) for k,v in l['paths'].items()}); print(l['drop_in_breakdown'])"
echo "== torchrun, 4 ranks on one GPU (gloo rehearsal) =="
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 10 --warmup 2 --backend gloo > gpurun_out/r04/bench_torchrun4_gloo.jsonl 2> gpurun_out/r04/bench_torchrun4_gloo.err; echo "rc=$?"; python -c "
import json
l=json.loads([x for x in open('gpurun_out/r04/bench_torchrun4_gloo.jsonl') if x.startswith('{')][-1])
print('value_from', l['value_from'], 'ms_per_step', l['ms_per_step']); print({k:(v.get('status'), v.get('ms_per_step')) for k,v in l['paths'].items()})"
echo "== config 5 counters =="
bash tools/gpu_profile_run.sh config5_streamed "--scene synthetic-100k --spp 64 --steps 2 --warmup 1" || exit 1
exit 0
