#!/bin/bash
# Round 4, visit R: the register budget of the 1-4 sphere builds (7 waves per SIMD since round 2) for this round's NEW builds of
# them — with a plane, with the general camera: 6 and 8 waves against 7.
set -o pipefail
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
cat > /tmp/tilt.py <<'PY'
PY
for scene in basic_plane basic; do
  echo "== $scene 1920x1080x256 =="
  timeout -k 10 500 python tools/gpu_ab.py $scene 1920 1080 256 15 librt_hip.so librt_hip_few6.so librt_hip_few8.so || exit 1
done 2>&1 | tee gpurun_out/r04/waves_few_ab.txt
