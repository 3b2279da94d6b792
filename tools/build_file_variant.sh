#!/bin/bash
# tools/build_file_variant.sh <name> <source stem> <flags...>: library variant with ONE translation unit rebuilt with extra flags
# (the other objects come from uncrtaints_amd/lib/obj)  ->  uncrtaints_amd/lib/ablate/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; stem=$2; shift 2
mkdir -p uncrtaints_amd/lib/ablate /tmp/fvar_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c uncrtaints_amd/csrc/$stem.hip -o /tmp/fvar_$name/$stem.o
objs=""
for o in uncrtaints_amd/lib/obj/*.o; do b=$(basename $o); if [ -f /tmp/fvar_$name/$b ]; then objs="$objs /tmp/fvar_$name/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o uncrtaints_amd/lib/ablate/lib_$name.so $objs
echo uncrtaints_amd/lib/ablate/lib_$name.so
