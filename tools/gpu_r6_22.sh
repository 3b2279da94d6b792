timeout 1500 python -m pytest tests/test_anysize.py tests/test_gpu_model.py -x -q -m gpu -k "anysize or any_size or odd or fresh or maxpool or small or thread or flat" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_variants.py -x -q -m gpu -k "small_input or instance_norm or multi_layer" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-bf16-leg --size 250 --steps 50 > gpurun_out/r06b_bench_fp32_250x250.json 2> gpurun_out/r06b_250.err; python -c "
import json
d=json.loads(open('gpurun_out/r06b_bench_fp32_250x250.json').read().strip().splitlines()[-1]); print('250x250', d['ms_per_step']); print(d['ltae_stage']['launches'])"
python bench.py --no-cpu-baseline --no-bf16-leg --steps 50 --no-kernel-events | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256', d['ms_per_step'])"
