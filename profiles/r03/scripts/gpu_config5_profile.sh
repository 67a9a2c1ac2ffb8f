#!/bin/bash
# BASELINE config 5 (synthetic 100k spheres, 1920x1080, 64 spp; the streamed kernel) under the profiler:
#   1. clocks and power around consecutive launches (sysfs sampler), back to back and with idle gaps
#   2. rocprofv3 --kernel-trace --stats
#   3. --pmc passes, each in its own run: SQ instruction mix, SQ wait / scalar cache, FETCH_SIZE, WRITE_SIZE
# Output under gpurun_out/c5/ (tools/summarise_profile.py condenses prof-style trees; this one is read by hand).
set -o pipefail
out=gpurun_out/c5
mkdir -p $out
export TMPDIR=/tmp
ARGS="bench.py --scene synthetic-100k --spp 64 --steps 1 --warmup 0 --settle-ms 0 --cpu-baseline-seconds 0 --no-kernel-only"
echo "== clocks: 4 launches back to back =="
timeout -k 10 200 python3 tools/gpu_clock_sampler.py $out/clocks_back_to_back.csv 100 -- python3 tools/gpu_config5_launches.py 4 0 > $out/launches_back_to_back.txt 2>&1 || { tail -20 $out/launches_back_to_back.txt; exit 1; }
cat $out/launches_back_to_back.txt | grep launch
echo "== clocks: 3 launches with 5 s idle between =="
timeout -k 10 200 python3 tools/gpu_clock_sampler.py $out/clocks_with_gaps.csv 100 -- python3 tools/gpu_config5_launches.py 3 5 > $out/launches_with_gaps.txt 2>&1 || { tail -20 $out/launches_with_gaps.txt; exit 1; }
cat $out/launches_with_gaps.txt | grep launch
echo "== kernel trace =="
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ARGS > $out/trace.log 2>&1 || { tail -20 $out/trace.log; exit 1; }
tail -1 $out/trace.log | cut -c1-400
pass() {
  name=$1; shift
  echo "== pmc $name: $* =="
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 $ARGS > $out/pmc_$name.log 2>&1 || { tail -20 $out/pmc_$name.log; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU && \
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR && \
pass sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_REQ_READ_16 && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE && \
pass tcc TCC_HIT_sum TCC_MISS_sum
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
