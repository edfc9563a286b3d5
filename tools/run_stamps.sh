#!/bin/bash
# phase timeline of the <16> conv32 kernels (needs the diagnostic build); restores the normal build afterwards
cd "$(dirname "$0")/.."
touch ar-vae_amd/csrc/conv32.hip
ARVAE_HIPCC_FLAGS="-DARVAE_STAMPS" python ar-vae_amd/build.py > /dev/null 2>&1
python tools/stamp_conv32.py ${1:-up} 2>&1 | tail -30
touch ar-vae_amd/csrc/conv32.hip; python ar-vae_amd/build.py > /dev/null 2>&1
