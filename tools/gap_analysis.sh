#!/bin/bash
# tools/gap_analysis.sh <tag>  ->  gpurun_out/<tag>_gaps.json : idle time between the kernels of the captured step (run through gpurun)
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gap_$tag
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_$tag -o gap -- \
    python "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-kernel-events > /tmp/gap_$tag.json 2> /tmp/gap_$tag.err </dev/null
python "$root/tools/gap_analysis.py" /tmp/gap_$tag "$root/gpurun_out/${tag}_gaps.json" 10
