#!/bin/bash
# per-step timeline of wgrad32r (diagnostic library tools/bin/lib_wgrst.so: conv32r.hip built with -DWGR_STAMPS): [lo size] [batch]
cd "$(dirname "$0")/.."
cp ar-vae_amd/libarvae_hip.so /tmp/lib_keep.so
cp tools/bin/lib_wgrst.so ar-vae_amd/libarvae_hip.so
python tools/stamp_wgr.py ${1:-16} ${2:-512} 2>/dev/null
cp /tmp/lib_keep.so ar-vae_amd/libarvae_hip.so
