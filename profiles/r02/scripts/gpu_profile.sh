#!/bin/bash
# rocprofv3 passes over bench.py (N=1): kernel trace + stats, then PMC counters in their own runs.
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --no-kernel-only $BENCH_ARGS"
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1 || true
echo "== the same command without the profiler (this box, for comparison) =="
timeout -k 10 300 python3 $ARGS > gpurun_out/prof/plain.log 2>&1 || { tail -20 gpurun_out/prof/plain.log; exit 1; }
tail -1 gpurun_out/prof/plain.log | cut -c1-300
echo "== kernel trace =="
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 $ARGS > gpurun_out/prof/trace.log 2>&1 || { tail -20 gpurun_out/prof/trace.log; exit 1; }
tail -2 gpurun_out/prof/trace.log
pass() {
  name=$1; shift
  echo "== pmc $name: $* =="
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/pmc_$name -- python3 $ARGS > gpurun_out/prof/pmc_$name.log 2>&1 || { tail -20 gpurun_out/prof/pmc_$name.log; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU && \
pass sq2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE && \
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
find gpurun_out/prof -name "*.csv" | head -40
