#!/bin/bash
# same-box sweep of the workgroup split of the four paired backward launches (per cent of the workgroups on the data gradient)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
{
echo "default            $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
for v in 16U 16D 8U 8D; do
  for sp in ${SPLITS:-40 44 54 58 62}; do
    echo "SPLIT$v $sp  $(env ARVAE_PAIR_SPLIT$v=$sp python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  done
done
echo "default            $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
} > gpurun_out/split_sweep.txt 2>&1
