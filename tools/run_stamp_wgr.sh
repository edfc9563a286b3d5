#!/bin/bash
cd "$(dirname "$0")/.."
touch ar-vae_amd/csrc/conv32r.hip
ARVAE_HIPCC_FLAGS="-DWGR_STAMPS $1" python ar-vae_amd/build.py > /dev/null 2>&1
python tools/stamp_wgr.py 16 ${2:-512} 2>/dev/null
touch ar-vae_amd/csrc/conv32r.hip; python ar-vae_amd/build.py > /dev/null 2>&1
