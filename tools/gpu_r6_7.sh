for c in 240 303 336 220; do python tools/fuzz_configs.py 1 $c --wide 2>&1 | grep -v amdgpu.ids | tail -6; done
python tools/fuzz_configs.py 1 26 2>&1 | grep -v amdgpu.ids | tail -5
timeout 900 python -m pytest tests/test_variants.py tests/test_anysize.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5
