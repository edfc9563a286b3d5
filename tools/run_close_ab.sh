#!/bin/bash
# round 6: the closing pair (dense_wgrad_batch + slab_reduce_batch as one grid) against the two launches back to back; diagnostic library both ways
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "image_step or baseline_batch_512 or full_batch or ragged or mnist" 2>&1 | tail -4 > gpurun_out/close_tests.txt
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2 3; do
  echo "A (paired close)  $(python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
  echo "B (back to back)  $(ARVAE_NO_PAIR_CLOSE=1 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | q)"
done > gpurun_out/close_ab.txt 2>&1
for w in mnist; do
  echo "A mnist $(python bench.py --no-cpu-baseline --workload $w --steps 30 2>/dev/null | q)"
  echo "B mnist $(ARVAE_NO_PAIR_CLOSE=1 python bench.py --no-cpu-baseline --workload $w --steps 30 2>/dev/null | q)"
done >> gpurun_out/close_ab.txt 2>&1
python bench.py --no-cpu-baseline --no-secondary --breakdown 2> gpurun_out/close_breakdown.txt > /dev/null
cat gpurun_out/close_tests.txt gpurun_out/close_ab.txt; grep -v amdgpu.ids gpurun_out/close_breakdown.txt
