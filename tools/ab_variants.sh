#!/bin/bash
# A/B of library variants inside ONE GPU session (run-to-run noise between boxes is ~1 %, and the first run of a session is
# slow): tools/ab_variants.sh <variant>...   -- variants are uncrtaints_amd/lib/ablate/lib_<variant>.so
# (tools/build_variant.sh); "base" = the shipped library.  One discarded warm-up run, then REPS interleaved rounds.
run() {
  if [ $1 = base ]; then L="X=1"; else L="UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$1.so"; fi
  env $L python bench.py --no-cpu-baseline --no-bf16-leg --no-kernel-events --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"
}
run base > /dev/null
for i in $(seq 1 ${REPS:-3}); do for v in base "$@"; do run $v; done; done
