#!/bin/bash
# one or more rocprofv3 --pmc passes over bench.py; usage: gpu_pmc.sh name "COUNTER COUNTER ..." [name "COUNTERS" ...]
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 $BENCH_ARGS"
while [ $# -ge 2 ]; do
  name=$1; counters=$2; shift 2
  rm -rf gpurun_out/prof/pmc_$name
  timeout -k 10 300 rocprofv3 --pmc $counters --output-format csv -d gpurun_out/prof/pmc_$name -- python3 $ARGS > gpurun_out/prof/pmc_$name.log 2>&1 || { tail -20 gpurun_out/prof/pmc_$name.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/prof/pmc_*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f.split('/')[2], k, len(v), sum(v)/len(v))
PY
