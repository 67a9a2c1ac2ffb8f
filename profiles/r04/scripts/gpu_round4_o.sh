#!/bin/bash
# Round 4, visit O: the new full-size plane-scene parity tests; the instruction-class passes for the other configurations.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== full-size plane scenes against the oracle =="
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "with_their_plane_at_full_size" 2>&1 | tail -5 || exit 1
MIX=1 bash tools/gpu_profile_run.sh config2_basic_64spp "--spp 64" || exit 1
MIX=1 bash tools/gpu_profile_run.sh config3_dielectric "--scene dielectric" || exit 1
MIX=1 bash tools/gpu_profile_run.sh config4_basic_4k "--width 3840 --height 2160" || exit 1
MIX=1 bash tools/gpu_profile_run.sh basic_plane_small "--scene basic_plane" || exit 1
MIX=1 bash tools/gpu_profile_run.sh dielectric_plane_small "--scene dielectric_plane" || exit 1
MIX=1 bash tools/gpu_profile_run.sh resident_64_spheres "--scene synthetic-64" || exit 1
exit 0
