timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/r06a_pytest_gpu.log | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/r06b_bench_fp32.json 2> gpurun_out/r06b_bench.err; tail -c 1500 gpurun_out/r06b_bench_fp32.json
