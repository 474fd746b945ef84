cd /tmp && export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --stats -d /root/repo/gpurun_out/prof_cfg5 -o t -- python /root/repo/tools/bench_cfg.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/prof_cfg5/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
