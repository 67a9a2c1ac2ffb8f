#!/bin/bash
# Round 5, visit P: rehearsals of the N > 1 plumbing on the one GPU a box has, with the bench.py of this round (first form's line at
# once, 300 s budget for the forms): four rank processes over gloo under torch.distributed.run as the driver starts it, and four
# members on one device behind one rt_hip_render (--same-device), gathered and direct-frame.  RCCL between distinct devices cannot
# be rehearsed here.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== torchrun, 4 ranks on one GPU (gloo rehearsal) =="
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 10 --warmup 2 --backend gloo --cpu-baseline-seconds 0 > gpurun_out/r05/p_bench_torchrun4_gloo.jsonl 2> gpurun_out/r05/p_bench_torchrun4_gloo.err; rc=$?; echo "rc=$rc"
python3 - <<'PY'
import json
for x in open('gpurun_out/r05/p_bench_torchrun4_gloo.jsonl'):
    if x.startswith('{'):
        l = json.loads(x)
        print(l.get('line'), l['n_gpus'], l['ms_per_step'], l['value'], {k: (v.get('status'), v.get('ms_per_step')) for k, v in (l.get('paths') or {}).items()})
PY
[ $rc -ne 0 ] && { tail -20 gpurun_out/r05/p_bench_torchrun4_gloo.err; exit $rc; }
echo "== one process, 4 members on one device =="
: > gpurun_out/r05/p_bench_multi_member.jsonl
for args in "--gpus 4 --same-device" "--gpus 4 --same-device --direct-frame" "--gpus 2 --same-device"; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --no-interactive --steps 10 --warmup 3 $args | tail -1 >> gpurun_out/r05/p_bench_multi_member.jsonl || exit 1
  tail -1 gpurun_out/r05/p_bench_multi_member.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['value'], d['config'].get('parallelism'), d.get('frame_matches_oracle'))"
done
