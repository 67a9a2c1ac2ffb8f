#!/bin/bash
# round 3, final GPU visit: the whole suite on the final code, every bench line, the rocprofv3 profiles
set -o pipefail
bash tools/gpu_round3_e.sh || exit $?
bash tools/gpu_round3_f.sh || exit $?
