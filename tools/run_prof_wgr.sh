#!/bin/bash
cd "$(dirname "$0")/.."
touch ar-vae_amd/csrc/conv32r.hip
ARVAE_HIPCC_FLAGS="$1" python ar-vae_amd/build.py > /dev/null 2>&1
tools/prof_link.sh wgrad 16 | head -2 | tail -1
tools/prof_link.sh wgrad 8 | head -2 | tail -1
touch ar-vae_amd/csrc/conv32r.hip; python ar-vae_amd/build.py > /dev/null 2>&1
