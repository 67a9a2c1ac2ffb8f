#!/bin/bash
# Round 4, visit K: counter passes (traffic, instruction mix) of the remaining BASELINE configurations with the final kernels,
# so that every configuration's bench line carries a measured roofline.traffic / roofline.issue.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
bash tools/gpu_profile_run.sh config2_basic_64spp "--spp 64" || exit 1
bash tools/gpu_profile_run.sh config3_dielectric "--scene dielectric" || exit 1
bash tools/gpu_profile_run.sh config4_basic_4k "--width 3840 --height 2160" || exit 1
bash tools/gpu_profile_run.sh dielectric_plane_small "--scene dielectric_plane" || exit 1
exit 0
