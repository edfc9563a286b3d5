#!/bin/bash
# round 6 (NOT built into the library: tools/patches/measure_wgrad_lane.patch adds it and its switches): MeasureVAE's queued weight
# gradients on a second stream beside the backward chain against the one-tail form; result in profiles/r6_lane_ab.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
{
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2; do
  echo "B (one tail)            $(ARVAE_MEASURE_NO_LANE=1 python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A (lane, 3 instalments) $(python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A mask 4 (enc upper)    $(ARVAE_MEASURE_LANE_MASK=4 python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A mask 2 (pre-encoder)  $(ARVAE_MEASURE_LANE_MASK=2 python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A mask 0 (tail on lane) $(ARVAE_MEASURE_LANE_MASK=0 python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A 192 CUs               $(ARVAE_MEASURE_LANE_CUS=192 python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A 128 CUs               $(ARVAE_MEASURE_LANE_CUS=128 python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A 192 CUs mask 4        $(ARVAE_MEASURE_LANE_CUS=192 ARVAE_MEASURE_LANE_MASK=4 python bench.py --no-cpu-baseline --workload measure 2>/dev/null | q)"
  echo "A graphs                $(python bench.py --no-cpu-baseline --workload measure --graphs 2>/dev/null | q)"
  echo "B graphs                $(ARVAE_MEASURE_NO_LANE=1 python bench.py --no-cpu-baseline --workload measure --graphs 2>/dev/null | q)"
done
} > gpurun_out/lane_ab.txt 2>&1
cat gpurun_out/lane_ab.txt
