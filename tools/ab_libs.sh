#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_libs.sh <a.so> <b.so> [bench args...]; alternates a, b, a, b
# (the builds are selected through ARVAE_LIB; the product library ar-vae_amd/libarvae_hip.so is never overwritten)
set -euo pipefail
cd "$(dirname "$0")/.."
a=$1; b=$2; shift 2
for v in $a $b; do test -f $v || { echo "$v is missing" >&2; exit 1; }; done
for rep in 1 2; do
  for v in $a $b; do
    echo "== $v: $(ARVAE_LIB=$PWD/$v python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), round(d["ms_per_step"],4))')"
  done
done
