#!/bin/bash
# A/B of library variants inside ONE GPU session (run-to-run noise between boxes is ~1 %): tools/ab_variants.sh <variant>...
# variants are uncrtaints_amd/lib/ablate/lib_<variant>.so (tools/build_variant.sh); "base" = the shipped library
for i in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then L="X=1"; else L="UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$v.so"; fi
  env $L python bench.py --no-cpu-baseline --no-kernel-events --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done
