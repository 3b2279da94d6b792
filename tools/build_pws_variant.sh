#!/bin/bash
# Variant of the library in which ONLY the split-GEMM objects are rebuilt with extra flags (fast: 5 objects, the rest is taken
# from uncrtaints_amd/lib/obj):  tools/build_pws_variant.sh <name> <flags...>  ->  uncrtaints_amd/lib/ablate/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p uncrtaints_amd/lib/ablate /tmp/pwsvar_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $*"
for p in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $F -DPWS_PRO=$p -c uncrtaints_amd/csrc/pw_gemm_split.hip -o /tmp/pwsvar_$name/pw_gemm_split_p$p.o &
done
wait
objs=""
for o in uncrtaints_amd/lib/obj/*.o; do
  b=$(basename $o)
  if [ -f /tmp/pwsvar_$name/$b ]; then objs="$objs /tmp/pwsvar_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o uncrtaints_amd/lib/ablate/lib_$name.so $objs
echo uncrtaints_amd/lib/ablate/lib_$name.so
