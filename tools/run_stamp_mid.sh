#!/bin/bash
cd "$(dirname "$0")/.."
touch ar-vae_amd/csrc/midblock.hip
ARVAE_HIPCC_FLAGS="-DMID_STAMPS $1" python ar-vae_amd/build.py > /dev/null 2>&1
python tools/stamp_mid.py 2>/dev/null
touch ar-vae_amd/csrc/midblock.hip; python ar-vae_amd/build.py > /dev/null 2>&1
