#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh <a.so> <b.so> [bench args...]; alternates a, b, a, b
cd "$(dirname "$0")/.."
a=$1; b=$2; shift 2
cp ar-vae_amd/libarvae_hip.so /tmp/lib_keep.so
for rep in 1 2; do
  for v in $a $b; do
    cp $v ar-vae_amd/libarvae_hip.so
    echo "== $v: $(python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],4))')"
  done
done
cp /tmp/lib_keep.so ar-vae_amd/libarvae_hip.so
