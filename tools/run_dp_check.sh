#!/bin/bash
# data-parallel code path on one GPU: its tests, then same-box step times of plain / forced-DP (overlap on, off) runs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/dp_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4), d["config"].get("launch",""))'; }
for rep in 1 2; do
  echo "plain        $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "dp overlap   $(python bench.py --no-cpu-baseline --no-secondary --force-dp 2>/dev/null | q)"
  echo "dp no-ovl    $(ARVAE_DP_OVERLAP=0 python bench.py --no-cpu-baseline --no-secondary --force-dp 2>/dev/null | q)"
done > gpurun_out/dp_bench.txt 2>&1
for rep in 1; do
  echo "measure plain graphs  $(python bench.py --workload measure --no-cpu-baseline 2>/dev/null | q)"
  echo "measure dp graphs     $(python bench.py --workload measure --no-cpu-baseline --force-dp 2>/dev/null | q)"
  echo "measure dp eager      $(python bench.py --workload measure --no-cpu-baseline --force-dp --no-graphs 2>/dev/null | q)"
done >> gpurun_out/dp_bench.txt 2>&1
