#!/bin/bash
# parity lines of a few tests under two builds of the library (GPU box): tools/ab_parity_lib.sh <variant .so> "<pytest -k expr>"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for tag in shipped variant; do
  L=""; [ $tag = variant ] && L=$PWD/$1
  UNCR_HIP_LIB=$L python -m pytest tests/test_gpu_model.py tests/test_variants.py -m gpu -q -s -k "$2" 2>&1 | grep "^\[parity\]" > gpurun_out/ab/$tag.txt
  echo "== $tag: $(wc -l < gpurun_out/ab/$tag.txt) lines"
  python - $tag <<'PY'
import re,sys
rows=[]
for l in open(f'gpurun_out/ab/{sys.argv[1]}.txt'):
    m=re.match(r"\[parity\] (\S+): vs fp32 ref (\S+); vs fp64 truth: hip (\S+), cpu-fp32 (\S+)", l)
    if m and 'grad[' in m.group(1): rows.append((float(m.group(3)), float(m.group(4)), m.group(1)))
rows.sort(reverse=True)
for a,b,k in rows[:10]: print(f"   hip {a:.2e} cpu {b:.2e} ratio {a/max(b,1e-12):5.2f}  {k}")
PY
done
