#!/bin/bash
# fresh-seed sweeps on the frozen sources (scratch script)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
f() { grep -v "^\[parity\]\|amdgpu.ids\|UserWarning\|run_backward\|^  File\|^    \|Traceback"; }
timeout 900 python tools/fuzz_configs.py 60 500 --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs_c.log; tail -1 gpurun_out/r06_fuzz_configs_c.log
timeout 900 python tools/fuzz_configs.py 50 700 --wide --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs_wide_c.log; tail -1 gpurun_out/r06_fuzz_configs_wide_c.log
timeout 400 python tools/fuzz_standalone.py 40 100 2>&1 | f > gpurun_out/r06_fuzz_standalone_c.log; tail -1 gpurun_out/r06_fuzz_standalone_c.log
timeout 400 python tools/fuzz_metrics.py 40 100 2>&1 | f > gpurun_out/r06_fuzz_metrics_c.log; tail -1 gpurun_out/r06_fuzz_metrics_c.log
timeout 400 python tools/fuzz_bf16.py 40 100 2>&1 | f > gpurun_out/r06_fuzz_bf16_c.log; tail -1 gpurun_out/r06_fuzz_bf16_c.log
