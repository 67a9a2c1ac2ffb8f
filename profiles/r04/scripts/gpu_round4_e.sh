#!/bin/bash
# Round 4, visit D2: A/B series.  (1) lane modes as scalar masks (librt_hip_masks.so) vs the mode register (the product) (the one
# structural experiment); (2) one 16-byte store/load per parked value vs the 8 + 4 byte pair, and the memory-model
# ordering of the arrival vs the ISA-level one; (3) config 5's counters (WRITE_SIZE).
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
ab() { out=$1; shift; echo "== $* ==" | tee -a gpurun_out/r04/$out; timeout -k 10 600 python tools/gpu_ab.py "$@" 2>&1 | tee -a gpurun_out/r04/$out; }
: > gpurun_out/r04/ab_mode_masks.txt; : > gpurun_out/r04/arrival_ordering_ab.txt; : > gpurun_out/r04/ab_publish_16_bytes.txt
ab ab_mode_masks.txt basic 1920 1080 256 30 librt_hip.so librt_hip_masks.so
ab ab_mode_masks.txt dielectric 1920 1080 256 30 librt_hip.so librt_hip_masks.so
ab ab_mode_masks.txt basic 1920 1080 64 40 librt_hip.so librt_hip_masks.so
ab ab_mode_masks.txt basic_plane 1920 1080 256 30 librt_hip.so librt_hip_masks.so
ab ab_mode_masks.txt synthetic-64 1920 1080 64 20 librt_hip.so librt_hip_masks.so
ab arrival_ordering_ab.txt synthetic-10000 1920 1080 32 5 librt_hip_split.so librt_hip_model.so librt_hip.so
ab arrival_ordering_ab.txt synthetic-2000 1920 1080 64 5 librt_hip_split.so librt_hip_model.so librt_hip.so
ab ab_publish_16_bytes.txt synthetic-100k 1920 1080 64 2 librt_hip_split.so librt_hip.so
echo "== why is the traced headline kernel slower than the untraced one? (carrier threads vs the profiler) =="
for v in "default::" "no_helpers:RT_HIP_COPY_THREADS=0:" "locked::--locked-frame"; do
  name=${v%%:*}; rest=${v#*:}; envs=${rest%%:*}; extra=${rest#*:}
  mkdir -p /tmp/tr_$name
  env $envs timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --no-kernel-only $extra > /tmp/tr_$name/plain.jsonl 2>/dev/null
  if [ -n "$envs" ]; then export $envs; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$name/trace -- python3 bench.py --steps 5 --warmup 2 --cpu-baseline-seconds 0 --no-kernel-only $extra > /tmp/tr_$name/traced.jsonl 2>/dev/null
  if [ -n "$envs" ]; then unset ${envs%%=*}; fi
  python3 - "$name" <<'PY' | tee -a gpurun_out/r04/traced_vs_plain.txt
import json, sys, glob, csv
name = sys.argv[1]
plain = json.loads(open(f"/tmp/tr_{name}/plain.jsonl").read().strip().splitlines()[-1])
traced = json.loads(open(f"/tmp/tr_{name}/traced.jsonl").read().strip().splitlines()[-1])
durs = []
for f in glob.glob(f"/tmp/tr_{name}/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "render_queue" in r["Kernel_Name"]:
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
print(f"{name:12s} plain: wall {plain['ms_per_step']:.4f} kernel(events) {plain['roofline']['kernel_ms']:.4f} | traced: wall {traced['ms_per_step']:.4f} kernel(events) {traced['roofline']['kernel_ms']:.4f} kernel(trace, last 5) {sum(durs[-5:])/5:.4f} (all {len(durs)}: min {min(durs):.4f} max {max(durs):.4f})")
PY
done
echo "== config 5 counters =="
bash tools/gpu_profile_run.sh config5_streamed "--scene synthetic-100k --spp 64 --steps 2 --warmup 1" || exit 1
exit 0
