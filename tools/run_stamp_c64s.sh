#!/bin/bash
# phase stamps of conv64s (diagnostic library tools/bin/lib_c64st.so, built with -DC64S_STAMPS)
cd "$(dirname "$0")/.."
cp ar-vae_amd/libarvae_hip.so /tmp/lib_keep.so
cp tools/bin/lib_c64st.so ar-vae_amd/libarvae_hip.so
python tools/stamp_c64s.py 2>&1 | tail -12
cp /tmp/lib_keep.so ar-vae_amd/libarvae_hip.so
