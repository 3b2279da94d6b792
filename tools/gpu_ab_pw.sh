#!/bin/bash
# tools/gpu_ab_pw.sh "<only list>" <variant>...   (run on the GPU box): base + variants, two rounds, kernels in isolation
cd $GRAFT_REPO_ROOT
only=$1; shift
python tools/bench_pw.py --only $only > /dev/null 2>&1
for r in 1 2; do
  python tools/bench_pw.py --only $only 2>&1 | tail -8
  for v in "$@"; do UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$v.so python tools/bench_pw.py --only $only 2>&1 | tail -8; done
done
