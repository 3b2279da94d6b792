run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --act-dtype bf16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', d['ms_per_step'])"; }
run X=1 base >/dev/null
for i in 1 2 3; do run X=1 base; run UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_nont35.so nont35; done
