#!/bin/bash
# tools/build_kres_variant.sh <name> <flags...>: library variant with only pw_gemm_kres.hip rebuilt -> uncrtaints_amd/lib/ablate/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p uncrtaints_amd/lib/ablate /tmp/krvar_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c uncrtaints_amd/csrc/experiments/pw_gemm_kres.hip -o /tmp/krvar_$name/pw_gemm_kres.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DUNCR_WITH_KRES -c uncrtaints_amd/csrc/pw_gemm.hip -o /tmp/krvar_$name/pw_gemm.o
objs=""
for o in uncrtaints_amd/lib/obj/*.o; do b=$(basename $o); if [ -f /tmp/krvar_$name/$b ]; then objs="$objs /tmp/krvar_$name/$b"; else objs="$objs $o"; fi; done
objs="$objs /tmp/krvar_$name/pw_gemm_kres.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o uncrtaints_amd/lib/ablate/lib_$name.so $objs
echo uncrtaints_amd/lib/ablate/lib_$name.so
