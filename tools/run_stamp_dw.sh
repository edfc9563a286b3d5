#!/bin/bash
cd "$(dirname "$0")/.."
touch ar-vae_amd/csrc/dense.hip
ARVAE_HIPCC_FLAGS="-DDW_STAMPS $1" python ar-vae_amd/build.py > /dev/null 2>&1
python tools/stamp_dw.py 2>/dev/null
touch ar-vae_amd/csrc/dense.hip; python ar-vae_amd/build.py > /dev/null 2>&1
