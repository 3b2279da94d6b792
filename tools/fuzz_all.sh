#!/bin/bash
# every fuzz sweep once, logs under gpurun_out/<tag>_fuzz_*.log (GPU box):  tools/fuzz_all.sh <tag>
tag=${1:-rXX}
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
f() { grep -v "^\[parity\]\|amdgpu.ids\|UserWarning\|run_backward\|^  File\|^    \|Traceback"; }
python tools/fuzz_configs.py 60 0 2>&1 | f > gpurun_out/${tag}_fuzz_configs.log; tail -1 gpurun_out/${tag}_fuzz_configs.log
python tools/fuzz_configs.py 40 200 --wide 2>&1 | f > gpurun_out/${tag}_fuzz_configs_wide.log; tail -1 gpurun_out/${tag}_fuzz_configs_wide.log
python tools/fuzz_standalone.py 40 0 2>&1 | f > gpurun_out/${tag}_fuzz_standalone.log; tail -1 gpurun_out/${tag}_fuzz_standalone.log
python tools/fuzz_metrics.py 40 0 2>&1 | f > gpurun_out/${tag}_fuzz_metrics.log; tail -1 gpurun_out/${tag}_fuzz_metrics.log
python tools/fuzz_bf16.py 40 0 2>&1 | f > gpurun_out/${tag}_fuzz_bf16.log; tail -1 gpurun_out/${tag}_fuzz_bf16.log
