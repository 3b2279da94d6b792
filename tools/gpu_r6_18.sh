timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/r06a_pytest_gpu.log | tail -15
REPS=3 bash tools/ab_variants.sh agg1 2>&1 | tee gpurun_out/r06_ab_agg1.log
