#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
f() { grep -v "^\[parity\]\|amdgpu.ids\|UserWarning\|run_backward\|^  File\|Traceback"; }
timeout 800 python tools/fuzz_configs.py 50 900 --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs_d.log; tail -1 gpurun_out/r06_fuzz_configs_d.log
timeout 800 python tools/fuzz_configs.py 50 1100 --wide --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs_wide_d.log; tail -1 gpurun_out/r06_fuzz_configs_wide_d.log
timeout 300 python tools/fuzz_standalone.py 40 200 2>&1 | f > gpurun_out/r06_fuzz_standalone_d.log; tail -1 gpurun_out/r06_fuzz_standalone_d.log
timeout 300 python tools/fuzz_bf16.py 30 200 2>&1 | f > gpurun_out/r06_fuzz_bf16_d.log; tail -1 gpurun_out/r06_fuzz_bf16_d.log
timeout 300 python tools/fuzz_small_ops.py 2>&1 | f | tail -5 > gpurun_out/r06_fuzz_small_ops_d.log; tail -1 gpurun_out/r06_fuzz_small_ops_d.log
