python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -s -k "two_part_weight_gradient or mbconv" 2>&1 | grep "parity\] weight\|passed\|failed\|Error\|assert" | head -30
run() { env $1 python bench.py --no-cpu-baseline --no-kernel-events --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run X=1 > /dev/null
for i in 1 2 3; do run X=1; run UNCR_NO_H2_DX=1; done
