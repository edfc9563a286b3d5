#!/bin/bash
# tools/build_diag.sh <out name> <source.hip> "<extra hipcc flags>": a diagnostic build of ONE source file linked with the
# product's other objects into tools/bin/<out name>.so (selected at run time with ARVAE_LIB; the product library is untouched)
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p tools/bin
python ar-vae_amd/build.py > /dev/null
src=ar-vae_amd/csrc/$2
obj=/tmp/diag_$(basename $2 .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $3 -c $src -o $obj
objs=$(ls ar-vae_amd/csrc/build/*.o | grep -v "/$(basename $2 .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/$1.so $objs $obj
echo "built tools/bin/$1.so"
