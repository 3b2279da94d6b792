#!/bin/bash
# tools/ablate_ltae_stage.sh <tag>: three interleaved (full, ablated) pairs per storage mode -> gpurun_out/<tag>_ltae_ablation.json
cd "$(dirname "$0")/.."
tag=${1:-rXX}; mkdir -p gpurun_out; out=gpurun_out/${tag}_ltae_ablation.jsonl; : > $out
python tools/ablate_ltae_stage.py > /dev/null 2>&1      # warm-up process
for m in "" bf16; do for i in 1 2 3; do
  python tools/ablate_ltae_stage.py $m 2>/dev/null | tail -1 >> $out
  python tools/ablate_ltae_stage.py --ablated $m 2>/dev/null | tail -1 >> $out
done; done
python - <<PY
import json
rows = [json.loads(l) for l in open("$out") if l.startswith("{")]
res = {}
for act in ("fp32", "bf16"):
    f = sorted(r["step_ms"] for r in rows if r["act_dtype"] == act and not r["ablated"])
    a = sorted(r["step_ms"] for r in rows if r["act_dtype"] == act and r["ablated"])
    if len(f) < 3 or len(a) < 3:
        continue
    d = f[1] - a[1]
    ab = rows[0]["algorithmic_bytes_of_the_stage"] // (1 if act == "fp32" else 2)
    res[act] = {"step_ms_full": f, "step_ms_without_stage_kernels": a, "stage_ms_by_ablation_incl_scatter_stats": round(d, 4),
                "algorithmic_bytes": ab, "roofline_frac_incl_scatter_stats": round(ab / (d * 1e-3) / 8e12, 4)}
res["method"] = ("median of three interleaved process pairs (each: median of 5 x 200 graph replays): the captured training step vs the same "
                 "step whose two L-TAE stage calls launch nothing and hand back an earlier step's results "
                 "(engine.dev_options(ltae_replay='replay')); the difference contains the pooled-gradient scatter + statistics pass, so the "
                 "like-for-like sum-of-kernels figure is bench.py's ltae_stage.roofline_frac_with_scatter_stats")
json.dump(res, open("gpurun_out/${tag}_ltae_ablation.json", "w"), indent=1)
print(json.dumps(res)[:900])
PY
