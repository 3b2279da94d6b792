timeout 900 python -m pytest tests/test_anysize.py -x -q -m gpu -k "flat_kernels or depthwise or maxpool" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_anysize.py -x -q -m gpu 2>&1 | tail -15
for c in 17 26 49 57; do python tools/fuzz_configs.py 1 $c 2>&1 | grep -v amdgpu.ids | tail -4; done
for c in 220 331 342 352; do python tools/fuzz_configs.py 1 $c --wide 2>&1 | grep -v amdgpu.ids | tail -4; done
