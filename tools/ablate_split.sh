#!/bin/bash
# Build ablated variants of the split GEMM (development aid): lib/ablate/libuncr_<mask>.so
# usage: [EXTRA=-DPWS_STAMP] tools/ablate_split.sh <mask>...
set -e
cd "$(dirname "$0")/.."
mkdir -p uncrtaints_amd/lib/ablate
OBJ=uncrtaints_amd/lib/obj
for m in "$@"; do
  for p in 0 1 2 3 4; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA -DPWS_ABL=$m -DPWS_PRO=$p -c uncrtaints_amd/csrc/pw_gemm_split.hip -o /tmp/pws_${m}_$p.o &
  done
  wait
  objs=$(ls $OBJ/*.o | grep -v pw_gemm_split)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o uncrtaints_amd/lib/ablate/libuncr_$m.so $objs /tmp/pws_${m}_[0-4].o
done
