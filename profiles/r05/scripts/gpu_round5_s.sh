#!/bin/bash
# Round 5, visit S: where the LDS-resident kernel should change from the LDS scan to the scalar-load scan (the threshold of 40 spheres
# dates from before the round's cuts; visit R's table jumps from 0.63 of the peak at 32 spheres to 0.69 at 40): builds with the
# threshold at 9 and at 20 against the shipped one, 9 to 40 spheres, 1080p x 64 spp, kernel ms; and bench.py under torchrun with one
# rank — its line now carries frame_matches_oracle for the frame the ranks assembled.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{
for n in 9 12 16 20 24 32 40; do
  echo "== synthetic-$n 1920 1080 64 =="; timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 10 librt_hip.so librt_hip_scalar9.so librt_hip_scalar20.so || exit 1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/s_resident_scan_threshold_ab.txt
echo "== bench.py under torchrun, one rank =="
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/r05/s_bench_torchrun1.jsonl 2> gpurun_out/r05/s_bench_torchrun1.err; rc=$?
python3 - <<'PY'
import json
for x in open('gpurun_out/r05/s_bench_torchrun1.jsonl'):
    if x.startswith('{'):
        l = json.loads(x); print(l.get('line', '')[:20], l['ms_per_step'], l['value'], l.get('frame_matches_oracle'), l.get('frame_sha256'), {k: v.get('status') for k, v in (l.get('paths') or {}).items()})
PY
echo "== torchrun, 4 ranks on one GPU (gloo rehearsal) =="
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 10 --warmup 2 --backend gloo --cpu-baseline-seconds 0 > gpurun_out/r05/s_bench_torchrun4_gloo.jsonl 2> gpurun_out/r05/s_bench_torchrun4_gloo.err; rc2=$?
python3 - <<'PY'
import json
for x in open('gpurun_out/r05/s_bench_torchrun4_gloo.jsonl'):
    if x.startswith('{'):
        l = json.loads(x); print(l.get('line', '')[:20], l['n_gpus'], l['ms_per_step'], l['value'], l.get('frame_matches_oracle'), {k: v.get('status') for k, v in (l.get('paths') or {}).items()})
PY
[ $rc -ne 0 ] && exit $rc
exit $rc2
