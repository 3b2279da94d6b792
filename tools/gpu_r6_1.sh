set -x
UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_wgstamp.so python tools/stamp_wgrad.py > gpurun_out/r06_stamp_wgrad.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06a_bench_fp32.json 2> gpurun_out/r06a_bench.err
tail -c 600 gpurun_out/r06a_bench_fp32.json
cat gpurun_out/r06_stamp_wgrad.log
