tools/round_evidence.sh r06
tools/gap_analysis.sh r06
python bench.py --no-cpu-baseline --no-bf16-leg --size 250 > gpurun_out/r06_bench_fp32_250x250.json 2> gpurun_out/r06_bench_250.err; tail -c 300 gpurun_out/r06_bench_fp32_250x250.json
f() { grep -v "^\[parity\]\|amdgpu.ids\|UserWarning\|run_backward\|^  File\|^    \|Traceback"; }
python tools/fuzz_standalone.py 40 0 2>&1 | f > gpurun_out/r06_fuzz_standalone.log; tail -1 gpurun_out/r06_fuzz_standalone.log
python tools/fuzz_metrics.py 40 0 2>&1 | f > gpurun_out/r06_fuzz_metrics.log; tail -1 gpurun_out/r06_fuzz_metrics.log
python tools/fuzz_bf16.py 40 0 2>&1 | f > gpurun_out/r06_fuzz_bf16.log; tail -1 gpurun_out/r06_fuzz_bf16.log
python tools/fuzz_configs.py 60 0 --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs.log; tail -1 gpurun_out/r06_fuzz_configs.log
python tools/fuzz_configs.py 60 200 --wide --pin-inconv 2>&1 | f > gpurun_out/r06_fuzz_configs_wide.log; tail -1 gpurun_out/r06_fuzz_configs_wide.log
