#!/bin/bash
# phase stamps of the latent block's forward launch (diagnostic library tools/bin/lib_midst.so: midblock.hip built with -DMID_STAMPS)
set -euo pipefail
cd "$(dirname "$0")/.."
test -f tools/bin/lib_midst.so || { echo "tools/bin/lib_midst.so is missing: build the diagnostic library first (tools/README.md)" >&2; exit 1; }
export ARVAE_LIB=$PWD/tools/bin/lib_midst.so          # _lib.py loads this build instead of the product library
python tools/stamp_mid.py 2>/dev/null
