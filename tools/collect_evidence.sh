#!/bin/bash
# copy the summaries of tools/round_evidence.sh <tag> from gpurun_out/ into profiles/ (tracked)
set -eu
tag=$1
cd "$(dirname "$0")/.."
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/${tag}_pytest_gpu.log | sed -e 's/\x1b\[[0-9;]*m//g' > profiles/${tag}_pytest_gpu.log
for f in parity stream_roofs power power_bf16; do [ -f gpurun_out/${tag}_$f.json ] && cp gpurun_out/${tag}_$f.json profiles/${tag}_$f.json; done
cp gpurun_out/${tag}_traffic.json profiles/${tag}_traffic.json
cp gpurun_out/${tag}_traffic_bf16.json profiles/${tag}_traffic_bf16.json
cp gpurun_out/${tag}_pmc_FETCH_SIZE.csv gpurun_out/${tag}_pmc_WRITE_SIZE.csv profiles/
cp gpurun_out/${tag}_pmc_bf16_FETCH_SIZE.csv gpurun_out/${tag}_pmc_bf16_WRITE_SIZE.csv profiles/
cp gpurun_out/${tag}_pipes.json profiles/${tag}_pipes.json
cp gpurun_out/${tag}_bench_fp32.json profiles/${tag}_bench_fp32.json
cp gpurun_out/${tag}_bench_bf16.json profiles/${tag}_bench_bf16.json
cp gpurun_out/${tag}_fp32_kernel_stats.csv profiles/${tag}_kernel_stats_fp32.csv
cp gpurun_out/${tag}_bf16_kernel_stats.csv profiles/${tag}_kernel_stats_bf16.csv
cp gpurun_out/${tag}_launches.json profiles/${tag}_launches.json
ls -la profiles | grep " ${tag}_"
