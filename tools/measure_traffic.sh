#!/bin/bash
# PMC HBM traffic of the step's dominant kernels (run on the GPU box through gpurun): counters in their own passes, with
# --kernel-trace only.   tools/measure_traffic.sh <tag> [bf16]  ->  gpurun_out/<tag>_traffic[_bf16].json (+ the two counter CSVs)
set -u
tag=${1:-r01}
mode=traffic; sfx=""; flag=""
if [ "${2:-}" = "bf16" ]; then mode=traffic_bf16; sfx="_bf16"; flag="--bf16"; fi
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python "$root/tools/bench_kernels.py" $mode > /tmp/pmc_$c.log 2>&1 </dev/null
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "no counter csv for $c"; tail -5 /tmp/pmc_$c.log; exit 1; fi
  cp "$f" "$root/gpurun_out/${tag}_pmc${sfx}_$c.csv"
done
python "$root/tools/parse_traffic.py" $flag "$root/gpurun_out/${tag}_pmc${sfx}_FETCH_SIZE.csv" "$root/gpurun_out/${tag}_pmc${sfx}_WRITE_SIZE.csv" "$root/gpurun_out/${tag}_traffic${sfx}.json"
