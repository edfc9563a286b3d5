#!/bin/bash
# build conv32r.hip variants and time wgrad<16> with each:  tools/ablate_wgr.sh   (run on the GPU box)
cd "$(dirname "$0")/.."
for flags in "" "-DWGR_NO_MFMA" "-DWGR_NO_PRODUCE" "-DWGR_NO_COMMIT" "-DWGR_NO_MFMA -DWGR_NO_COMMIT"; do
  touch ar-vae_amd/csrc/conv32r.hip
  ARVAE_HIPCC_FLAGS="$flags" python ar-vae_amd/build.py > /dev/null 2>&1
  echo "flags: '$flags'"; python tools/time_link.py wgrad 16 2>/dev/null; python tools/time_link.py wgrad 8 2>/dev/null
done
touch ar-vae_amd/csrc/conv32r.hip; python ar-vae_amd/build.py > /dev/null 2>&1
