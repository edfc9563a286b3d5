#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > gpurun_out/b2_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4), d["config"].get("launch",""))'; }
for rep in 1 2; do
  echo "pair prep    $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "no pair prep $(ARVAE_NO_PAIR_PREP=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/b2_bench.txt 2>&1
echo "measure dp graphs     $(python bench.py --workload measure --no-cpu-baseline --force-dp 2>gpurun_out/b2_mdp.err | q)" >> gpurun_out/b2_bench.txt 2>&1
bash tools/trace_kernels.sh all BENCH_ARGS=--force-dp > gpurun_out/b2_trace_dp_overlap.txt 2>&1
bash tools/trace_kernels.sh all BENCH_ARGS=--force-dp ARVAE_DP_OVERLAP=0 > gpurun_out/b2_trace_dp_plain.txt 2>&1
