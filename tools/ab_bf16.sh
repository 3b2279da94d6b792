#!/bin/bash
# interleaved A/B of library variants on the bf16 bench (development): tools/ab_bf16.sh <name>... ; "base" = the in-tree library
cd "$(dirname "$0")/.."
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --act-dtype bf16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'], ' '.join('%s=%.4f' % (r['kernel'].split(',N')[0][:24]+(',N12' if ',N12' in r['kernel'] else ''), r['mean_ms']) for r in d['kernel_breakdown'] if r['kernel'].startswith(('pw_gemm','dw_'))))"; }
for rep in 1 2; do for v in "$@"; do
  echo -n "$v: "
  if [ "$v" = base ]; then run; elif [ "${v#slots}" != "$v" ]; then UNCR_PWS_SLOTS=${v#slots} run; else UNCR_HIP_LIB=uncrtaints_amd/lib/ablate/lib_$v.so run; fi
done; done
