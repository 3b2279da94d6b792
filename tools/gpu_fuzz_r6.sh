python tools/fuzz_configs.py 60 0 --pin-inconv 2>&1 | grep -v "amdgpu.ids\|^\[parity\]" > gpurun_out/r06_fuzz_configs.log; tail -2 gpurun_out/r06_fuzz_configs.log
python tools/fuzz_configs.py 60 200 --wide --pin-inconv 2>&1 | grep -v "amdgpu.ids\|^\[parity\]" > gpurun_out/r06_fuzz_configs_wide.log; tail -2 gpurun_out/r06_fuzz_configs_wide.log
for c in 303 331 336 342 352; do python tools/fuzz_configs.py 1 $c --wide --pin-inconv 2>&1 | grep "^case\|^      "; done > gpurun_out/r06_fuzz_flagged_round5.log; cat gpurun_out/r06_fuzz_flagged_round5.log
