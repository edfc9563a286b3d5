#!/bin/bash
# phase stamps of the clustered latent block (diagnostic library tools/bin/lib_midcst.so: midcluster.hip built with -DMIDC_STAMPS;
# `bash tools/build_diag.sh lib_midcst midcluster.hip -DMIDC_STAMPS` builds it)
set -euo pipefail
cd "$(dirname "$0")/.."
test -f tools/bin/lib_midcst.so || bash tools/build_diag.sh lib_midcst midcluster.hip -DMIDC_STAMPS
export ARVAE_LIB=$PWD/tools/bin/lib_midcst.so          # _lib.py loads this build instead of the product library
python tools/stamp_midc.py 2>/dev/null
