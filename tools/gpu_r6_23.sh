UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_trot1.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pw_gemm or mbconv or split" 2>&1 | tail -3
REPS=4 bash tools/ab_variants.sh trot1 2>&1 | tee gpurun_out/r06_ab_trot1.log
timeout 600 python -m pytest tests/test_anysize.py -x -q -m gpu -k "maxpool" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-bf16-leg --size 250 --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('250x250', d['ms_per_step']); print(d['ltae_stage']['launches']['fwd'])"
