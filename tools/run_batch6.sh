#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/b6_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4), d["config"].get("launch",""))'; }
for rep in 1 2; do
  echo "down32p      $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "down32k      $(ARVAE_D32K_NO_PC=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/b6_bench.txt 2>&1
bash tools/trace_kernels.sh down32 > gpurun_out/b6_trace.txt 2>&1
bash tools/trace_kernels.sh down32 ARVAE_D32K_NO_PC=1 >> gpurun_out/b6_trace.txt 2>&1
