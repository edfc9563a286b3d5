#!/bin/bash
# phase stamps of the rows GEMM (diagnostic library tools/bin/lib_rgst.so, dense.hip built with -DRG_STAMPS)
set -euo pipefail
cd "$(dirname "$0")/.."
test -f tools/bin/lib_rgst.so || { echo "tools/bin/lib_rgst.so is missing: build the diagnostic library first (tools/README.md)" >&2; exit 1; }
export ARVAE_LIB=$PWD/tools/bin/lib_rgst.so          # _lib.py loads this build instead of the product library
python tools/stamp_rg.py 2>&1 | tail -20
