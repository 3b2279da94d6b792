set -x
UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_wgstamp.so python tools/stamp_wgrad.py > gpurun_out/r06_stamp_wgrad.log 2>&1
cat gpurun_out/r06_stamp_wgrad.log
UNCR_HIP_LIB=$PWD/uncrtaints_amd/lib/ablate/lib_map1.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pw_gemm or mbconv or split" 2>&1 | tail -5
REPS=3 bash tools/ab_variants.sh map1 2>&1 | tee gpurun_out/r06_ab_map1.log
