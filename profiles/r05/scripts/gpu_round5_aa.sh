#!/bin/bash
# Round 5, visit AA: the LDS-resident kernel built per camera form (pinhole / the others) and per scan (LDS copy / scalar loads)
# instead of one kernel that carries all of it: its loop reloaded the frame's constants from the argument block (30 s_load, 35
# v_readlane of spilled scalars per loop body) — librt_hip_resforms.so against the shipped library, 1080p x 64 spp, kernel ms; parity first.
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "== parity of the experiment build: the resident kernel's tests =="
RT_HIP_LIBRARY=rt_amd/lib/librt_hip_resforms.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "resident or mid_sizes or random_scenes or orthographic or varying_w or frame_is_bit_exact" > gpurun_out/r05/aa_pytest.txt 2>&1; rc=$?; tail -3 gpurun_out/r05/aa_pytest.txt
[ $rc -ne 0 ] && exit $rc
{
for n in 9 12 24 32 40 64 200 1000; do
  echo "== synthetic-$n 1920 1080 64 =="; timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 6 librt_hip.so librt_hip_resforms.so || exit 1
done
for n in 12 64; do
  echo "== synthetic-$n 1920 1080 64, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py synthetic-$n 1920 1080 64 6 librt_hip.so librt_hip_resforms.so || exit 1
done
echo "== basic 1920 1080 256, resident forced =="; AB_FLAGS=2 timeout -k 10 300 python tools/gpu_ab.py basic 1920 1080 256 6 librt_hip.so librt_hip_resforms.so || exit 1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/aa_resident_forms_ab.txt
