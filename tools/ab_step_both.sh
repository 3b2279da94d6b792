#!/bin/bash
# tools/ab_step_both.sh <variant>: interleaved A/B of the in-tree library against lib_<variant>.so on the fp32 and the bf16 step
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16-leg --no-power --no-kernel-events $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'])"; }
V=$1
run X=1 warm > /dev/null
for i in 1 2 3 4; do run X=1 base_fp32; run UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$V.so ${V}_fp32; done
for i in 1 2 3; do run X=1 base_bf16 "--act-dtype bf16"; run UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_$V.so ${V}_bf16 "--act-dtype bf16"; done
