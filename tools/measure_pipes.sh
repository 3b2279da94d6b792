#!/bin/bash
# Where the cycles of the step's dominant kernels go (run on the GPU box through gpurun): two SQ counter passes over
# `tools/bench_kernels.py traffic`, counters in their own runs with --kernel-trace only.
#   tools/measure_pipes.sh <tag>  ->  gpurun_out/<tag>_pipes.json (+ the raw counter CSVs)
set -u
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
i=0
for set in "$A" "$B"; do
  i=$((i+1))
  rm -rf /tmp/pipes_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pipes_$i -o p -- python "$root/tools/bench_kernels.py" traffic > /tmp/pipes_$i.log 2>&1 </dev/null
  f=$(find /tmp/pipes_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "no counter csv for pass $i"; tail -5 /tmp/pipes_$i.log; continue; fi
  cp "$f" "$root/gpurun_out/${tag}_pipes_pass$i.csv"
done
python "$root/tools/parse_pipes.py" "$root/gpurun_out/${tag}_pipes.json" "$root"/gpurun_out/${tag}_pipes_pass*.csv
