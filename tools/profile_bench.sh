#!/bin/bash
# rocprofv3 per-kernel summary of the default bench step (run on the GPU box through gpurun):
#   tools/profile_bench.sh <tag> [bench.py args...]   ->  gpurun_out/<tag>_kernel_stats.csv + gpurun_out/<tag>_bench.json
set -u
tag=${1:-prof}; shift || true
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- \
    python "$root/bench.py" --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-bf16-leg "$@" > "$root/gpurun_out/${tag}_bench.json" 2> /tmp/prof_$tag.err </dev/null
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then
    cp "$f" "$root/gpurun_out/${tag}_kernel_stats.csv"
    head -45 "$f" | cut -c1-260
else
    echo "no kernel_stats.csv produced"; tail -20 /tmp/prof_$tag.err
fi
