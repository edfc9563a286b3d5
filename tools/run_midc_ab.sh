#!/bin/bash
# round 5: what the ticketed places of the clustered latent block cost (diagnostic library, ARVAE_MIDC_STATIC = places by blockIdx)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export ARVAE_LIB=$PWD/ar-vae_amd/libarvae_hip_diag.so
bash tools/trace_kernels.sh midc_ > gpurun_out/midc_tickets.txt 2>&1
bash tools/trace_kernels.sh midc_ ARVAE_MIDC_STATIC=1 > gpurun_out/midc_static.txt 2>&1
bash tools/trace_kernels.sh midc_ > gpurun_out/midc_tickets2.txt 2>&1
bash tools/trace_kernels.sh midc_ ARVAE_MIDC_STATIC=1 > gpurun_out/midc_static2.txt 2>&1
tail -3 gpurun_out/midc_tickets.txt gpurun_out/midc_static.txt gpurun_out/midc_tickets2.txt gpurun_out/midc_static2.txt
