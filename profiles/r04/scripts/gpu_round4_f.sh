#!/bin/bash
# Round 4, visit F: the GPU suite ONCE with planes' hopeless-lane skip, the general-camera build of the scalar-register
# kernels and the persistent launches capped at 5 workgroups per CU; then the A/Bs these changes ask for.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== pytest -m gpu (once) =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 500 -p no:cacheprovider > gpurun_out/r04/pytest_gpu.txt 2>&1; rc=$?; tail -15 gpurun_out/r04/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
ab() { out=$1; shift; echo "== $* ==" | tee -a gpurun_out/r04/$out; timeout -k 10 600 python tools/gpu_ab.py "$@" 2>&1 | tee -a gpurun_out/r04/$out; }
: > gpurun_out/r04/persistent_waves_ab.txt
ab persistent_waves_ab.txt synthetic-100k 1920 1080 64 2 librt_hip_waves6.so librt_hip.so
ab persistent_waves_ab.txt synthetic-10000 1920 1080 32 5 librt_hip_waves6.so librt_hip.so
ab persistent_waves_ab.txt synthetic-2000 1920 1080 64 5 librt_hip_waves6.so librt_hip.so
echo "== bench lines: plane scenes, tilted camera =="
: > gpurun_out/r04/bench_planes.jsonl
for a in "--scene basic" "--scene basic_plane" "--scene basic_plane --resident" "--scene basic --resident" "--scene dielectric" "--scene dielectric_plane" "--scene dielectric_plane --resident" "--scene basic --tilt" "--scene basic --tilt --resident" "--scene basic_plane --tilt" "--scene synthetic-64 --spp 64" "--scene synthetic-64"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0 $a >> gpurun_out/r04/bench_planes.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
  tail -1 gpurun_out/r04/bench_planes.jsonl | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$a', l['ms_per_step'], 'ms', l['roofline']['kernel'], l['roofline']['kernel_ms'], 'frac', l['roofline']['frac'])"
done
exit 0
