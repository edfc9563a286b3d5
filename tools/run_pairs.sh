#!/bin/bash
# parity tests, then same-box comparisons of the paired data-gradient / weight-gradient launches of the 16x16 and 8x8 layers:
# default split, ARVAE_NO_PAIR32=1 (two launches per layer), and the share of workgroups given to the data gradient
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/pairs_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
{
for rep in 1 2; do
  echo "paired (default)  $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "two launches      $(ARVAE_NO_PAIR32=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done
for sp in ${SPLITS:-40 44 48 52}; do
  echo "split16 $sp  $(ARVAE_PAIR_SPLIT16=$sp python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "split8  $sp  $(ARVAE_PAIR_SPLIT8=$sp python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done
} > gpurun_out/pairs_bench.txt 2>&1
bash tools/trace_kernels.sh all > gpurun_out/pairs_trace.txt 2>&1
