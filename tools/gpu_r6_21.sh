timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "inconv" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_variants.py tests/test_anysize.py -x -q -m gpu 2>&1 | tail -6
for c in 26 ; do python tools/fuzz_configs.py 1 $c --pin-inconv 2>&1 | grep "^case\|^      "; done
for c in 220 247 258; do python tools/fuzz_configs.py 1 $c --wide --pin-inconv 2>&1 | grep "^case\|^      "; done
