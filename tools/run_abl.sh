#!/bin/bash
# diagnostic: phase stamps of up32<16> for ablation builds in tools/bin/lib_*.so
cd "$(dirname "$0")/.."
cp ar-vae_amd/libarvae_hip.so /tmp/lib_keep.so
for V in "$@"; do
  cp tools/bin/lib_$V.so ar-vae_amd/libarvae_hip.so
  echo "#### $V"
  python tools/stamp_conv32.py 512 2>&1 | sed -n '/== up32/,/== wgrad32/p' | grep -v wgrad32 | cut -c1-200
  echo "#### $V gated"
  ARVAE_STAMP_GATE=1 python tools/stamp_conv32.py 512 2>&1 | sed -n '/== up32/,/== wgrad32/p' | grep -E "whole|tile 1|drain" | cut -c1-200
done
cp /tmp/lib_keep.so ar-vae_amd/libarvae_hip.so
