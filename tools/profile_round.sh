#!/bin/bash
# the round's evidence in one GPU call: rocprofv3 kernel stats + FETCH / WRITE passes of the headline bench (tools/profile.sh), the bench
# lines of the four workloads, kernel stats of cfg5 / cfg3 / cfg4.  usage: tools/profile_round.sh <tag>  -> gpurun_out/round_<tag>/
TAG=${1:-rXX}
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/round_$TAG
mkdir -p "$OUT"
bash tools/profile.sh $TAG > "$OUT/profile.log" 2>&1
cp gpurun_out/prof_$TAG/summary.txt "$OUT/summary.txt"
cp gpurun_out/prof_$TAG/trace/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || find gpurun_out/prof_$TAG/trace -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
python bench.py 2>/dev/null | tail -1 > "$OUT/bench_line.json"
python bench.py --workload cfg5 2>/dev/null | tail -1 > "$OUT/bench_cfg5.json"
python bench.py --workload cfg3 2>/dev/null | tail -1 > "$OUT/bench_cfg3.json"
python bench.py --workload cfg4 2>/dev/null | tail -1 > "$OUT/bench_cfg4.json"
export TMPDIR=/tmp
for w in cfg5 cfg3 cfg4; do
  ( cd /tmp && rocprofv3 -f csv --kernel-trace --stats -d "$OUT/trace_$w" -o t -- python "$ROOT/bench.py" --workload $w --steps 20 --warmup 5 > /dev/null 2>&1 )
  python - "$OUT" $w <<'PY'
import csv, glob, sys
out, w = sys.argv[1], sys.argv[2]
f = glob.glob(f'{out}/trace_{w}/**/*kernel_stats.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(f'{out}/{w}_kernel_summary.txt', 'w') as fh:
        for r in rows[:16]:
            fh.write(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}\n")
PY
  rm -rf "$OUT/trace_$w"
done
ls -la "$OUT"
