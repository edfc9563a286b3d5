#!/bin/bash
# phase stamps of the rows GEMM (diagnostic library tools/bin/lib_rgst.so, dense.hip built with -DRG_STAMPS)
cd "$(dirname "$0")/.."
cp ar-vae_amd/libarvae_hip.so /tmp/lib_keep.so
cp tools/bin/lib_rgst.so ar-vae_amd/libarvae_hip.so
python tools/stamp_rg.py 2>&1 | tail -20
cp /tmp/lib_keep.so ar-vae_amd/libarvae_hip.so
