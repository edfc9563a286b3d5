#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/b5_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4), d["config"].get("launch",""))'; }
{
echo "dsprites     $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
echo "measure      $(python bench.py --workload measure --no-cpu-baseline 2>/dev/null | q)"
echo "measure eager $(python bench.py --workload measure --no-cpu-baseline --no-graphs 2>/dev/null | q)"
echo "mnist        $(python bench.py --workload mnist --no-cpu-baseline 2>/dev/null | q)"
} > gpurun_out/b5_bench.txt 2>&1
