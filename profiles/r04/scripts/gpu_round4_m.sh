#!/bin/bash
# Round 4, visit M: the helpers' wake-up moved behind the launch — A/B against the build before, helpers hot and asleep.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
nproc; python - <<'PY'
import os
print('cpus allowed', len(os.sched_getaffinity(0)))
try:
    print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except OSError as e:
    print('cpu.max', e)
PY
timeout -k 10 900 python tools/gpu_carrier_wake_ab.py 60 > gpurun_out/r04/carrier_wake_ab.txt 2>&1 || { tail -20 gpurun_out/r04/carrier_wake_ab.txt; exit 1; }
cat gpurun_out/r04/carrier_wake_ab.txt
