#!/bin/bash
# phase stamps of conv64s (diagnostic library tools/bin/lib_c64st.so, built with -DC64S_STAMPS)
set -euo pipefail
cd "$(dirname "$0")/.."
test -f tools/bin/lib_c64st.so || { echo "tools/bin/lib_c64st.so is missing: build the diagnostic library first (tools/README.md)" >&2; exit 1; }
export ARVAE_LIB=$PWD/tools/bin/lib_c64st.so          # _lib.py loads this build instead of the product library
python tools/stamp_c64s.py 2>&1 | tail -12
