#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (one rocprofv3 --pmc run each, nothing else traced) for the BASELINE configurations other
# than the headline: what feeds `roofline.traffic` of their bench lines (profiles/pmc_traffic.json).
set -o pipefail
export TMPDIR=/tmp
run() {
  tag=$1; shift
  mkdir -p gpurun_out/prof_$tag
  for counter in FETCH_SIZE WRITE_SIZE; do
    name=$(echo $counter | tr 'A-Z' 'a-z' | sed 's/_size//')
    rm -rf gpurun_out/prof_$tag/pmc_$name
    timeout -k 10 300 rocprofv3 --pmc $counter --output-format csv -d gpurun_out/prof_$tag/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline-seconds 0 --no-kernel-only "$@" > gpurun_out/prof_$tag/pmc_$name.log 2>&1 || { tail -5 gpurun_out/prof_$tag/pmc_$name.log; return 1; }
  done
  echo "traffic passes of $tag done"
}
run c2 --spp 64 && run c3 --scene dielectric && run c4 --width 3840 --height 2160
