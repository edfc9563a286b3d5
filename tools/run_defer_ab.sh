#!/bin/bash
# round 6: the forward pass's finishing step carried by the backward pass's first launch (ARVAE_VAE_DEFER_FINISH) against its own launch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py tests/test_inference_gpu.py tests/test_data_gpu.py -q -m gpu -k "not measure" 2>&1 | tail -8 > gpurun_out/defer_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2 3; do
  echo "A (deferred finish)  $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "B (own launch)       $(ARVAE_DEFER_FINISH=0 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/defer_ab.txt 2>&1
cat gpurun_out/defer_tests.txt gpurun_out/defer_ab.txt
