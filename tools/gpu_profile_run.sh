#!/bin/bash
# usage: tools/gpu_profile_run.sh TAG "bench.py arguments"      (on the GPU box, inside one gpurun call)
# rocprofv3 over ONE bench.py command: plain run, --kernel-trace --stats, then PMC counters in passes of their own
# (never together with a trace: MI355X_MICROARCH.md / gpurun's rule).  Summaries land under gpurun_out/$ROUND/TAG/ (ROUND defaults to r05) —
# kernel_stats.csv, render_kernel_durations_ms.txt, pmc_summary.csv, counters.json (stamped with the kernel sources' hash,
# merged into profiles/pmc_counters.json at home by tools/merge_counters.py).
set -o pipefail
TAG=$1; BENCH_ARGS=$2
OUT=gpurun_out/${ROUND:-r05}/$TAG; RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --no-kernel-only $BENCH_ARGS"
echo "== [$TAG] plain: python3 $ARGS =="
timeout -k 10 400 python3 $ARGS > $OUT/bench_plain.jsonl 2> $RAW/plain.err || { tail -20 $RAW/plain.err; exit 1; }
cut -c1-260 $OUT/bench_plain.jsonl
echo "== [$TAG] kernel trace =="
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- python3 $ARGS > $OUT/bench_traced.jsonl 2> $RAW/trace.err || { tail -20 $RAW/trace.err; exit 1; }
pass() {
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $RAW/pmc_$name -- python3 $ARGS > $RAW/pmc_$name.log 2>&1 || { tail -20 $RAW/pmc_$name.log; return 1; }
}
echo "== [$TAG] pmc passes =="
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU && \
pass sq2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE || exit 1
if [ -n "$MIX" ]; then   # the vector instruction mix by class (what the stream can issue at: profiles/r04/valu_issue_costs.txt)
  pass mix1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 && \
  pass mix2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS || exit 1
fi
python3 tools/summarise_profile_r4.py "$TAG" "$RAW" "$OUT" "python3 $ARGS"
