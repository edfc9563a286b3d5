#!/bin/bash
# tools/run_ab.sh "<ENV=1 for variant B>" [pytest -k expression]: parity tests, then three same-box A/B rounds of the dSprites bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -x -q -m gpu ${2:+-k "$2"} 2>&1 | tail -6 > gpurun_out/ab_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2 3; do
  echo "A (default)  $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "B ($1)  $(env $1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/ab_bench.txt 2>&1
bash tools/trace_kernels.sh all > gpurun_out/ab_trace.txt 2>&1
