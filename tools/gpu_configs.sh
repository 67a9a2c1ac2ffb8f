#!/bin/bash
# bench lines for the BASELINE.json configurations that fit one GPU (config 4's frame is rendered on one GPU here)
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/configs.jsonl
: > $out
run() { echo "== $*"; timeout -k 10 900 python bench.py --cpu-baseline-seconds 0 "$@" | tail -1 >> $out || exit 1; tail -1 $out | cut -c1-330; }
run --steps 20 --warmup 3                                                     # headline: basic 1920x1080x256
run --steps 20 --warmup 3 --spp 64                                            # config 2: basic 1920x1080x64
run --steps 10 --warmup 2 --scene dielectric                                  # config 3: dielectric 1920x1080x256
run --steps 5 --warmup 1 --width 3840 --height 2160                           # config 4's frame on ONE gpu: basic 3840x2160x256
run --steps 1 --warmup 0 --settle-ms 0 --scene synthetic-100k --spp 64                      # config 5: synthetic 100k spheres 1920x1080x64
run --steps 20 --warmup 3 --width 256 --height 256 --spp 1                    # config 1's size (the reference's CPU-runnable case)
