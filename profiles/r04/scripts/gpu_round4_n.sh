#!/bin/bash
# Round 4, visit N: the vector instruction mix by class (SQ_INSTS_VALU_*), headline kernel and the streamed kernel of config 5.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
MIX=1 bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
grep -E "mix" gpurun_out/r04/headline_basic_1080p_256spp/pmc_summary.csv
MIX=1 bash tools/gpu_profile_run.sh config5_streamed "--scene synthetic-100k --spp 64 --steps 2 --warmup 1" || exit 1
grep -E "mix" gpurun_out/r04/config5_streamed/pmc_summary.csv
exit 0
