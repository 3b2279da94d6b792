#!/bin/bash
# tools/gpu_ab_pw_n.sh <N> "<only list>" <variant>...   kernels in isolation at N frames
cd $GRAFT_REPO_ROOT
N=$1; only=$2; shift 2
python tools/bench_pw.py --N $N --only $only > /dev/null 2>&1
for r in 1 2; do
  python tools/bench_pw.py --N $N --only $only 2>&1 | tail -8
  for v in "$@"; do UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$v.so python tools/bench_pw.py --N $N --only $only 2>&1 | tail -8; done
done
