#!/bin/bash
# Round 3, visit q: config 5 with one-sample items under the profiler (trace + PMC) and with per-wave clocks; the suite.
set -o pipefail
mkdir -p gpurun_out/q
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
out=gpurun_out/c5
rm -rf $out; mkdir -p $out
ARGS="bench.py --scene synthetic-100k --spp 64 --steps 1 --warmup 0 --settle-ms 0 --cpu-baseline-seconds 0 --no-kernel-only"
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_clocks.so timeout -k 10 200 python tools/gpu_wave_tail.py 2>&1 | grep -v amdgpu.ids > gpurun_out/q/wave_tail_sample_items.txt; echo "wave tail: rc $?" | tee gpurun_out/q/status.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $ARGS > $out/trace.log 2>&1; echo "trace: rc $?" | tee -a gpurun_out/q/status.txt
pass() {
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 $ARGS > $out/pmc_$name.log 2>&1 || { tail -5 $out/pmc_$name.log; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU && \
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE
echo "pmc: rc $?" | tee -a gpurun_out/q/status.txt
cat gpurun_out/q/wave_tail_sample_items.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/q/pytest_gpu.txt 2>&1
echo "GPU suite: rc $?" | tee -a gpurun_out/q/status.txt
tail -3 gpurun_out/q/pytest_gpu.txt
timeout -k 10 400 bash tools/gpu_configs.sh > gpurun_out/q/configs.log 2>&1; cp gpurun_out/configs.jsonl gpurun_out/q/configs.jsonl
timeout -k 10 300 python bench.py > gpurun_out/q/bench_default.jsonl 2> gpurun_out/q/bench_default.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/q/*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["config"]["workload"][:42], j["n_gpus"], j["ms_per_step"], j["value"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], (j.get("kernel_only") or {}).get("ms_per_step"), (j.get("plug_in_call") or {}).get("ms_per_step"))
PY
