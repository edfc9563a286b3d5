#!/bin/bash
# phase stamps of the latent block's forward launch (diagnostic library tools/bin/lib_midst.so: midblock.hip built with -DMID_STAMPS)
cd "$(dirname "$0")/.."
cp ar-vae_amd/libarvae_hip.so /tmp/lib_keep.so
cp tools/bin/lib_midst.so ar-vae_amd/libarvae_hip.so
python tools/stamp_mid.py 2>/dev/null
cp /tmp/lib_keep.so ar-vae_amd/libarvae_hip.so
