#!/bin/bash
# Everything the judged numbers come from, in one GPU session (run on the GPU box through gpurun, ~12 GPU-minutes):
#   tools/round_evidence.sh <tag>      e.g.  gpurun --timeout 2400 -- 'tools/round_evidence.sh r03'
# Writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (tools/collect_evidence.sh <tag>).
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
# PMC passes first: bench.py attaches roofline.traffic only from a file measured on the current kernel sources
tools/measure_traffic.sh $tag > gpurun_out/${tag}_traffic.log 2>&1; tail -9 gpurun_out/${tag}_traffic.log
tools/measure_traffic.sh $tag bf16 > gpurun_out/${tag}_traffic_bf16.log 2>&1; tail -9 gpurun_out/${tag}_traffic_bf16.log
mkdir -p profiles; cp gpurun_out/${tag}_traffic.json profiles/${tag}_traffic.json; cp gpurun_out/${tag}_traffic_bf16.json profiles/${tag}_traffic_bf16.json
tools/measure_pipes.sh $tag > gpurun_out/${tag}_pipes.log 2>&1
python bench.py > gpurun_out/${tag}_bench_fp32.json 2> gpurun_out/${tag}_bench_fp32.err
python bench.py --no-cpu-baseline --act-dtype bf16 > gpurun_out/${tag}_bench_bf16.json 2> gpurun_out/${tag}_bench_bf16.err
tools/profile_bench.sh ${tag}_fp32 > gpurun_out/${tag}_prof_fp32.log 2>&1
tools/profile_bench.sh ${tag}_bf16 --act-dtype bf16 > gpurun_out/${tag}_prof_bf16.log 2>&1
python - <<PY
import json
for f in ("gpurun_out/${tag}_bench_fp32.json", "gpurun_out/${tag}_bench_bf16.json"):
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["ms_per_step"], d["value"], r["kernel"], r["frac"], r["traffic"], d["ltae_stage"]["roofline_frac"], d.get("cpu_baseline", {}).get("value"))
PY
