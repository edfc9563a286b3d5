#!/bin/bash
# round-end check on the GPU box: the whole GPU suite, then the profile round (tools/profile_round.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/final_tests.txt
bash tools/profile_round.sh ${1:-r6}
