cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p250
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p250 -o p250 -- python bench.py --size 250 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --no-power --no-bf16-leg > gpurun_out/p250/bench.log 2>&1
tail -2 gpurun_out/p250/bench.log | cut -c1-400
find gpurun_out/p250 -name "*kernel_stats.csv" | head
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/p250/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:30]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.2f}")
PY
