#!/bin/bash
# round 5: what the ticketed places of the clustered latent block cost (diagnostic library): ticket heads 4 (default) / 8 / 1
# against places by blockIdx (ARVAE_MIDC_STATIC)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
for rep in 1 2; do
  for v in "X=1" "ARVAE_MIDC_STATIC=1" "ARVAE_MIDC_HEADS=8" "ARVAE_MIDC_HEADS=1"; do
    echo "[$v] $(bash tools/trace_kernels.sh midc_ $v 2>&1 | tail -1)"
  done
done
