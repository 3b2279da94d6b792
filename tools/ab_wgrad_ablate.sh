#!/bin/bash
for v in base wga1 wga2 wga4 wga3 wga5 wga6; do
  if [ $v = base ]; then L=""; else L="UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$v.so"; fi
  env $L python tools/bench_pw.py --N 4 --only wg1,wg2 2>&1 | grep wgrad | sed "s/^base/$v/"
done
