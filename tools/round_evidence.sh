#!/bin/bash
# Everything the judged numbers come from, in one GPU session (run on the GPU box through gpurun, ~12 GPU-minutes):
#   tools/round_evidence.sh <tag>      e.g.  gpurun --timeout 2400 -- 'tools/round_evidence.sh r03'
# Writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (tools/collect_evidence.sh <tag>).
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"; mkdir -p gpurun_out
# -s -v: the [parity] lines stay in the log; tools/parse_parity.py turns them into the tracked per-test margins
python -m pytest tests -m gpu -s -v -p no:cacheprovider > gpurun_out/${tag}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${tag}_pytest_gpu.log
python tools/parse_parity.py gpurun_out/${tag}_pytest_gpu.log gpurun_out/${tag}_parity.json | head -3
python tools/stream_roofs.py gpurun_out/${tag}_stream_roofs.json > gpurun_out/${tag}_stream_roofs.log 2>&1; tail -2 gpurun_out/${tag}_stream_roofs.log
python tools/clock_watch.py gpurun_out/${tag}_power.json > gpurun_out/${tag}_power.log 2>&1; tail -3 gpurun_out/${tag}_power.log
python tools/clock_watch.py gpurun_out/${tag}_power_bf16.json --bf16 > gpurun_out/${tag}_power_bf16.log 2>&1; tail -3 gpurun_out/${tag}_power_bf16.log
# PMC passes first: bench.py attaches roofline.traffic only from a file measured on the current kernel sources
tools/measure_traffic.sh $tag > gpurun_out/${tag}_traffic.log 2>&1; tail -9 gpurun_out/${tag}_traffic.log
tools/measure_traffic.sh $tag bf16 > gpurun_out/${tag}_traffic_bf16.log 2>&1; tail -9 gpurun_out/${tag}_traffic_bf16.log
mkdir -p profiles; cp gpurun_out/${tag}_traffic.json profiles/${tag}_traffic.json; cp gpurun_out/${tag}_traffic_bf16.json profiles/${tag}_traffic_bf16.json
tools/measure_pipes.sh $tag > gpurun_out/${tag}_pipes.log 2>&1
python bench.py > gpurun_out/${tag}_bench_fp32.json 2> gpurun_out/${tag}_bench_fp32.err
python bench.py --no-cpu-baseline --act-dtype bf16 > gpurun_out/${tag}_bench_bf16.json 2> gpurun_out/${tag}_bench_bf16.err
tools/profile_bench.sh ${tag}_fp32 > gpurun_out/${tag}_prof_fp32.log 2>&1
tools/profile_bench.sh ${tag}_bf16 --act-dtype bf16 > gpurun_out/${tag}_prof_bf16.log 2>&1
tools/launch_count.sh $tag
python - <<PY
import json
for f in ("gpurun_out/${tag}_bench_fp32.json", "gpurun_out/${tag}_bench_bf16.json"):
    d = json.load(open(f)); r = d["roofline"]
    print(f, d["ms_per_step"], d["value"], r["kernel"], r["frac"], r["traffic"], d["ltae_stage"]["roofline_frac"], d.get("cpu_baseline", {}).get("value"))
PY
