#!/bin/bash
# Scaling curve on ONE node: bench.py at 1 / 2 / 4 / 8 GPUs (one rank per GPU over RCCL/xGMI), one JSON line each.
#   tools/run_scale.sh [steps] [warmup] [extra bench.py flags...]     ->  gpurun_out/scale_N<k>.json + a summary table
# N = 1 runs bench.py directly; N > 1 through torch.distributed.run exactly as the driver launches it.  Needs N visible GPUs;
# sizes above the visible device count are skipped (and said so).  Fails fast: a run that errors stops the sweep.
set -u
steps=${1:-200}; warm=${2:-20}; shift 2 2>/dev/null || true
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
ndev=$(python -c "import torch; print(torch.cuda.device_count())")
port=29571
for n in 1 2 4 8; do
  if [ "$n" -gt "$ndev" ]; then echo "[scale] N=$n skipped: $ndev GPU(s) visible"; continue; fi
  out=gpurun_out/scale_N$n.json
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps $steps --warmup $warm --no-cpu-baseline "$@" > $out 2> gpurun_out/scale_N$n.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((port + n)) \
      bench.py --gpus $n --steps $steps --warmup $warm "$@" > $out 2> gpurun_out/scale_N$n.err
  fi
  rc=$?
  if [ $rc -ne 0 ] || [ ! -s $out ]; then echo "[scale] N=$n FAILED (rc $rc)"; tail -20 gpurun_out/scale_N$n.err; exit 1; fi
done
python - <<'PY'
import glob, json, re
rows = []
for f in sorted(glob.glob("gpurun_out/scale_N*.json"), key=lambda s: int(re.search(r"N(\d+)", s).group(1))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c = d.get("collective") or {}
    rows.append((d["n_gpus"], d["value"], d["ms_per_step"], d["step_hbm_roofline_frac"], c.get("wait_ms_per_step"), c.get("rccl_version")))
base = rows[0][1] / rows[0][0] if rows else 0
print("N  samples/s  ms/step  per-GPU roofline frac  scaling eff.  all-reduce wait ms  RCCL")
for n, v, ms, fr, w, ver in rows:
    print(f"{n:<2d} {v:9.1f}  {ms:7.3f}  {fr:21.4f}  {v / (n * base):12.3f}  {w if w is not None else '-':>18}  {ver or '-'}")
PY
