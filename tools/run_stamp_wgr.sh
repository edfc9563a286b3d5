#!/bin/bash
# per-step timeline of wgrad32r (diagnostic library tools/bin/lib_wgrst.so: conv32.hip built with -DWGR_STAMPS): [lo size] [batch]
set -euo pipefail
cd "$(dirname "$0")/.."
test -f tools/bin/lib_wgrst.so || { echo "tools/bin/lib_wgrst.so is missing: build the diagnostic library first (tools/README.md)" >&2; exit 1; }
export ARVAE_LIB=$PWD/tools/bin/lib_wgrst.so          # _lib.py loads this build instead of the product library
python tools/stamp_wgr.py ${1:-16} ${2:-512} 2>&1 | grep -v amdgpu.ids
