#!/bin/bash
# round 6: Morpho-MNIST's wide Linear layers on the tile GEMMs (dense.hip wide_gemm_x3_kernel) against the row kernels (ARVAE_MID_NO_WIDE=1, diagnostic library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py tests/test_inference_gpu.py -x -q -m gpu -k "mnist or dense" 2>&1 | tail -15 > gpurun_out/wide_tests.txt
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
q() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],4))'; }
for rep in 1 2; do
  echo "A (tile GEMMs)  $(python bench.py --no-cpu-baseline --workload mnist --steps 30 2>/dev/null | q)"
  echo "B (row kernels) $(ARVAE_MID_NO_WIDE=1 python bench.py --no-cpu-baseline --workload mnist --steps 30 2>/dev/null | q)"
done > gpurun_out/wide_ab.txt 2>&1
unset ARVAE_LIB
WORKLOAD=mnist bash tools/trace_kernels.sh all > gpurun_out/wide_trace.txt 2>&1
cat gpurun_out/wide_tests.txt gpurun_out/wide_ab.txt; grep -i "wide\|mid_\|dense_wgrad\|sum of" gpurun_out/wide_trace.txt
