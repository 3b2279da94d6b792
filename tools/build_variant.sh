#!/bin/bash
# Build a variant of the whole library with extra compiler flags (development aid):
#   tools/build_variant.sh <name> <flags...>  ->  uncrtaints_amd/lib/ablate/lib_<name>.so   (use with UNCR_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p uncrtaints_amd/lib/ablate /tmp/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC ${NOSLP--fno-slp-vectorize} $*"
for s in norm ew pw_gemm pw_wgrad_split pw_wgrad_a16 dwconv dwconv_row se ltae ltae_fused aggregate mgnll metrics conv3 attn_rows optim inconv anysize; do
  /opt/rocm/bin/hipcc $F -c uncrtaints_amd/csrc/$s.hip -o /tmp/var_$name/$s.o &
done
for p in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $F -DPWS_PRO=$p -c uncrtaints_amd/csrc/pw_gemm_split.hip -o /tmp/var_$name/pws_$p.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o uncrtaints_amd/lib/ablate/lib_$name.so /tmp/var_$name/*.o
echo uncrtaints_amd/lib/ablate/lib_$name.so
