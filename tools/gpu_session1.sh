set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/stream_roofs.py gpurun_out/r04_stream_roofs.json > gpurun_out/r04_stream_roofs.log 2>&1; tail -4 gpurun_out/r04_stream_roofs.log
tools/measure_pipes.sh r04_before > gpurun_out/r04_before_pipes.log 2>&1; tail -10 gpurun_out/r04_before_pipes.log
python -m pytest tests -m gpu -s -v -p no:cacheprovider > gpurun_out/r04_pytest_gpu_sv.log 2>&1; tail -3 gpurun_out/r04_pytest_gpu_sv.log
