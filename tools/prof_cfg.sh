# rocprofv3 kernel stats of tools/bench_cfg.py for one configuration (default: cfg5 on one GPU); summary -> profiles/<name>/
# usage: prof_cfg.sh [name-filter] [profile-dir-name]
filter=${1:-"32 pairs"}
name=${2:-cfg5}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_$name
rocprofv3 -f csv --kernel-trace --stats -d /root/repo/gpurun_out/prof_$name -o t -- python /root/repo/tools/bench_cfg.py "$filter" > /root/repo/gpurun_out/prof_$name.line 2>/dev/null
python - "$name" <<'PY'
import csv, glob, sys
name = sys.argv[1]
f = glob.glob(f'/root/repo/gpurun_out/prof_{name}/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = open(f'/root/repo/gpurun_out/prof_{name}.summary.txt', 'w')
for r in rows[:14]:
    line = f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}"
    print(line); out.write(line + "\n")
PY
cat /root/repo/gpurun_out/prof_$name.line
