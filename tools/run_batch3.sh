#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > gpurun_out/b3_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4), d["config"].get("launch",""))'; }
for rep in 1 2; do
  echo "pair reg     $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "no pair reg  $(ARVAE_NO_PAIR_REG=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/b3_bench.txt 2>&1
echo "dp (default) $(python bench.py --no-cpu-baseline --no-secondary --force-dp 2>/dev/null | q)" >> gpurun_out/b3_bench.txt
echo "measure dp graphs     $(python bench.py --workload measure --no-cpu-baseline --force-dp 2>gpurun_out/b3_mdp.err | q)" >> gpurun_out/b3_bench.txt 2>&1
bash tools/trace_kernels.sh all > gpurun_out/b3_trace.txt 2>&1
