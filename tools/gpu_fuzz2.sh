f() { grep -v "^\[parity\]\|amdgpu.ids\|UserWarning\|run_backward\|^  File\|^    \|Traceback"; }
python tools/fuzz_configs.py 60 60 --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs_b.log; tail -1 gpurun_out/r06_fuzz_configs_b.log
python tools/fuzz_configs.py 60 260 --wide --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs_wide_b.log; tail -1 gpurun_out/r06_fuzz_configs_wide_b.log
grep "<<<<<<" gpurun_out/r06_fuzz_configs_b.log gpurun_out/r06_fuzz_configs_wide_b.log | cut -c1-400
