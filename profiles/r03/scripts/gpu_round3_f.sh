#!/bin/bash
# round 3, GPU visit F: rocprofv3 of the final code — the headline command (kernel trace + PMC passes) and config 5 (kernel trace, instruction mix, scalar cache, FETCH/WRITE)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof gpurun_out/c5
bash tools/gpu_profile.sh > gpurun_out/profile_run.txt 2>&1 || { tail -30 gpurun_out/profile_run.txt; exit 1; }
tail -15 gpurun_out/profile_run.txt
out=gpurun_out/c5
mkdir -p $out
ARGS="bench.py --scene synthetic-100k --spp 64 --steps 1 --warmup 0 --settle-ms 0 --cpu-baseline-seconds 0 --no-kernel-only"
echo "== config 5: kernel trace =="
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ARGS > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
pass() {
  name=$1; shift
  echo "== config 5 pmc $name: $* =="
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 $ARGS > $out/pmc_$name.log 2>&1 || { tail -20 $out/pmc_$name.log; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU && \
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR && \
pass sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/c5/pmc_*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f.split('/')[2], k, len(v), sum(v)/len(v))
PY
