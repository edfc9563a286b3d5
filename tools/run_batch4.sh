#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/b4_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4), d["config"].get("launch",""))'; }
for rep in 1 2; do
  echo "batch wgrad  $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "no batch     $(ARVAE_NO_WGRAD_BATCH=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/b4_bench.txt 2>&1
echo "dp (default) $(python bench.py --no-cpu-baseline --no-secondary --force-dp 2>/dev/null | q)" >> gpurun_out/b4_bench.txt
bash tools/trace_kernels.sh all > gpurun_out/b4_trace.txt 2>&1
