#!/bin/bash
# tools/ab_devopt.sh <k=v[,k=v]> [reps]: interleaved A/B of the shipped path against `bench.py --dev-options <...>` inside one GPU
# session (fp32 step, then the bf16-storage step)
cd "$(dirname "$0")/.."
O=$1; R=${2:-3}
run() { python bench.py --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-bf16-leg --no-power --no-kernel-events $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
run warm > /dev/null
for i in $(seq 1 $R); do run base_fp32; run ${O}_fp32 "--dev-options $O"; done
for i in $(seq 1 $R); do run base_bf16 "--act-dtype bf16"; run ${O}_bf16 "--act-dtype bf16 --dev-options $O"; done
