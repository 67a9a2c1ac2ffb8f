#!/bin/bash
# Round 4, visit L (the tree as it will be judged): smoke, the GPU suite ONCE, the default bench line.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== smoke =="
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 || exit 1
echo "== pytest -m gpu (once) =="
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout 500 -p no:cacheprovider > gpurun_out/r04/pytest_gpu_final.txt 2>&1; rc=$?; tail -6 gpurun_out/r04/pytest_gpu_final.txt
[ $rc -ne 0 ] && exit $rc
echo "== the default bench line (what the driver runs) =="
timeout -k 10 300 python bench.py > gpurun_out/r04/bench_headline_final.jsonl 2>/tmp/bench.err || { tail -5 /tmp/bench.err; exit 1; }
python - <<'PY'
import json
l = json.loads([x for x in open('gpurun_out/r04/bench_headline_final.jsonl') if x.startswith('{')][-1])
r = l['roofline']
print('value', l['value'], l['unit'], 'ms_per_step', l['ms_per_step'], 'kernel', r['kernel_ms'], 'frac', r['frac'], 'traffic', r['traffic'])
print('issue', r.get('issue'))
print('cpu_baseline', l.get('cpu_baseline'))
print('other mode', l.get('other_frame_mode', {}).get('ms_per_step'), 'plug-in', l.get('plug_in_call', {}).get('ms_per_step'))
PY
exit 0
