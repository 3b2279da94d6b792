python tools/fuzz_configs.py 1 331 --wide --shape=2,5,64,32 2>&1 | grep -v amdgpu.ids | tail -4
python tools/fuzz_configs.py 1 331 --wide --shape=2,5,66,34 2>&1 | grep -v amdgpu.ids | tail -4
for c in 240 303 336 247 258; do python tools/fuzz_configs.py 1 $c --wide 2>&1 | grep -v amdgpu.ids | tail -8; done
timeout 600 python -m pytest tests/test_variants.py -x -q -m gpu -k "instance or norm" 2>&1 | tail -5
