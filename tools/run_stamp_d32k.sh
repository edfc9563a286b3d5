#!/bin/bash
cd "$(dirname "$0")/.."
touch ar-vae_amd/csrc/conv32k.hip
ARVAE_HIPCC_FLAGS="-DD32K_STAMPS $1" python ar-vae_amd/build.py > /dev/null 2>&1
ARVAE_D32K_ONLY16=1 python tools/stamp_d32k.py 2>/dev/null
touch ar-vae_amd/csrc/conv32k.hip; python ar-vae_amd/build.py > /dev/null 2>&1
