"""Idle time between consecutive kernels of the captured training step (GPU box): is launch latency hidden inside a HIP-graph replay?
    rocprofv3 --kernel-trace --output-format csv -d /tmp/gap -o gap -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-kernel-events
    python tools/gap_analysis.py /tmp/gap  ->  gpurun_out/<tag>_gaps.json (tools/gap_analysis.sh wraps both)
Takes the dispatches of the timed replays (the last K * L dispatches of the stream's most frequent step pattern), and reports per step:
kernel time, idle time between kernels (start[i+1] - end[i] > 0), the gap histogram and the gaps behind the launches under 30 us."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
if not f:
    sys.exit("no kernel_trace.csv under " + root)
rows = list(csv.DictReader(open(f[0])))
key = lambda r, *names: next(r[n] for n in names if n in r)
ev = sorted(((int(key(r, "Start_Timestamp", "Start")), int(key(r, "End_Timestamp", "End")), key(r, "Kernel_Name", "Name")) for r in rows))
# the step's first kernel: the weight-pack launch of the forward (one per step)
first = [i for i, e in enumerate(ev) if "pack_wt_batch" in e[2]]
if len(first) < steps + 1:
    sys.exit(f"only {len(first)} steps found")
bounds = first[-(steps + 1):]
res = []
for a, b in zip(bounds[:-1], bounds[1:]):
    seg = ev[a:b]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
    small = [g for g, (s, e, _) in zip(gaps, seg[:-1]) if e - s < 30000]
    res.append(dict(launches=len(seg), wall_us=(seg[-1][1] - seg[0][0]) / 1e3, kernel_us=busy / 1e3, idle_us=sum(gaps) / 1e3,
                    idle_behind_small_us=sum(small) / 1e3, small_launches=len(small), small_kernel_us=sum(e - s for s, e, _ in seg if e - s < 30000) / 1e3,
                    gap_hist_us={"<1": sum(g < 1000 for g in gaps), "1-2": sum(1000 <= g < 2000 for g in gaps),
                                 "2-5": sum(2000 <= g < 5000 for g in gaps), ">5": sum(g >= 5000 for g in gaps)},
                    max_gap_us=max(gaps) / 1e3))
med = lambda k: sorted(r[k] for r in res)[len(res) // 2]
summary = {k: med(k) for k in ("launches", "wall_us", "kernel_us", "idle_us", "idle_behind_small_us", "small_launches", "small_kernel_us", "max_gap_us")}
summary["gap_hist_us"] = res[len(res) // 2]["gap_hist_us"]
summary["note"] = ("median over %d graph replays; idle = sum of positive gaps between one kernel's end and the next kernel's start on the "
                   "device timeline (rocprofv3 kernel trace)" % len(res))
print(json.dumps(summary, indent=1))
if out:
    json.dump(dict(summary=summary, steps=res), open(out, "w"), indent=1)
