#!/bin/bash
# Round 5, visit N: the final kernels (sources 984e72e903153e41) on every frame of the round's A/B table; the round's soaks over the
# new arithmetic paths (contract v4 draws, eye form, one-plane rule): random scenes x kernel modes, rank shares, big-scene item forms;
# one bench.py line per BASELINE.json configuration, for the plane / tilted / 64-sphere / 12-sphere frames, and under torchrun with
# one rank (the N > 1 code path's plumbing as the driver starts it).
set -o pipefail
mkdir -p gpurun_out/r05
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
{
for cfg in "basic 1920 1080 256" "basic 1920 1080 64" "basic 3840 2160 256" "scenes/basic_plane.toml 1920 1080 256" "dielectric 1920 1080 256" "dielectric_plane 1920 1080 256" "synthetic-8 1920 1080 256" "synthetic-5 1920 1080 256" "synthetic-64 1920 1080 256" "synthetic-12 1920 1080 64" "basic 256 256 1"; do
  echo "== $cfg =="; timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so || exit 1
done
for cfg in "basic 1920 1080 256" "scenes/basic_plane.toml 1920 1080 256" "dielectric 1920 1080 256"; do
  echo "== $cfg, tilted camera =="; AB_TILT=1 timeout -k 10 300 python tools/gpu_ab.py $cfg 15 librt_hip.so || exit 1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/n_final_ab.txt
echo "== soak: random scenes x 11 modes =="
RT_HIP_RANDOM_CASES=3000 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k random_scenes > gpurun_out/r05/n_soak_random_scenes.txt 2>&1; rc=$?; tail -2 gpurun_out/r05/n_soak_random_scenes.txt
[ $rc -ne 0 ] && exit $rc
echo "== soak: rank shares of small scenes =="
timeout -k 10 900 python tools/gpu_partition_soak.py 4000 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r05/n_soak_partition.txt || exit 1
echo "== soak: big-scene item forms =="
timeout -k 10 900 python tools/gpu_big_scene_soak.py 300 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r05/n_soak_big_scenes.txt || exit 1
echo "== bench lines, all configurations =="
bash tools/gpu_configs.sh && cp gpurun_out/configs.jsonl gpurun_out/r05/n_bench_all_configs.jsonl || exit 1
echo "== bench lines: planes, tilt, 64 and 12 spheres =="
out=gpurun_out/r05/n_bench_more.jsonl; : > $out
for args in "--scene basic_plane" "--scene basic_plane --tilt" "--tilt" "--scene dielectric_plane" "--scene synthetic-64" "--scene synthetic-12 --spp 64"; do
  timeout -k 10 300 python bench.py --cpu-baseline-seconds 0 --no-interactive --steps 10 --warmup 3 $args | tail -1 >> $out || exit 1; tail -1 $out | cut -c1-200
done
echo "== bench.py under torchrun, one rank =="
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-baseline-seconds 0 > gpurun_out/r05/n_bench_torchrun1.jsonl 2> gpurun_out/r05/n_bench_torchrun1.err; rc=$?; cut -c1-300 gpurun_out/r05/n_bench_torchrun1.jsonl
exit $rc
