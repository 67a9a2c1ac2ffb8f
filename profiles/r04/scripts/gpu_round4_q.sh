#!/bin/bash
# Round 4, visit Q: every profiled configuration once more with the final kernels (the 7 + 1 build's register budget changed the
# kernel sources' hash, and bench.py drops counter figures measured on other sources): trace, PMC, instruction classes.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
export MIX=1
bash tools/gpu_profile_run.sh headline_basic_1080p_256spp "" || exit 1
bash tools/gpu_profile_run.sh config2_basic_64spp "--spp 64" || exit 1
bash tools/gpu_profile_run.sh config3_dielectric "--scene dielectric" || exit 1
bash tools/gpu_profile_run.sh config4_basic_4k "--width 3840 --height 2160" || exit 1
bash tools/gpu_profile_run.sh basic_plane_small "--scene basic_plane" || exit 1
bash tools/gpu_profile_run.sh basic_plane_resident "--scene basic_plane --resident" || exit 1
bash tools/gpu_profile_run.sh dielectric_plane_small "--scene dielectric_plane" || exit 1
bash tools/gpu_profile_run.sh basic_tilted_camera "--scene basic --tilt" || exit 1
bash tools/gpu_profile_run.sh resident_64_spheres "--scene synthetic-64" || exit 1
bash tools/gpu_profile_run.sh config5_streamed "--scene synthetic-100k --spp 64 --steps 2 --warmup 1" || exit 1
exit 0
