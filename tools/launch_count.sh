#!/bin/bash
# tools/launch_count.sh <tag>: kernel launches per training step and the time in launches under 30 us -> gpurun_out/<tag>_launches.json
# (run on the GPU box)
set -u
tag=${1:-rXX}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"; mkdir -p gpurun_out
# launches per step: two traces of the bare timed loop (no event-profiled re-run, no power-probe replays -- their step counts
# depend on wall time), 10 and 20 steps; the difference leaves the set-up launches (optimizer state, buffers) out
STEPS=10 tools/profile_bench.sh ${tag}_fp32x1 --no-power --no-kernel-events > gpurun_out/${tag}_prof_fp32x1.log 2>&1
STEPS=20 tools/profile_bench.sh ${tag}_fp32x2 --no-power --no-kernel-events > gpurun_out/${tag}_prof_fp32x2.log 2>&1
python - <<PY
import csv, json
def load(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("gpurun_out/${tag}_fp32x1_kernel_stats.csv"), load("gpurun_out/${tag}_fp32x2_kernel_stats.csv")
per = {k: ((b[k][0] - a.get(k, (0, 0))[0]) / 10.0, (b[k][1] - a.get(k, (0, 0))[1]) / 10.0 / 1e3) for k in b}
per = {k: v for k, v in per.items() if v[0] > 0}
small = {k: v for k, v in per.items() if v[1] / v[0] < 30.0}
out = {"launches_per_step": sum(v[0] for v in per.values()), "kernel_us_per_step": sum(v[1] for v in per.values()),
       "launches_under_30us": sum(v[0] for v in small.values()), "us_in_launches_under_30us": sum(v[1] for v in small.values()),
       "method": "rocprofv3 kernel stats of bench.py --no-power --no-kernel-events at 20 steps minus the same at 10 steps, divided by 10",
       "per_kernel": {k: [round(v[0], 2), round(v[1], 1)] for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
json.dump(out, open("gpurun_out/${tag}_launches.json", "w"), indent=1)
print("launches/step", out["launches_per_step"], "under 30 us:", out["launches_under_30us"], round(out["us_in_launches_under_30us"], 1), "us")
PY
