#!/bin/bash
# rocprofv3 evidence for the RSNT render (tools/bench_rsnt.py): per-kernel stats and, in a pass of its own, the MFMA counters of the
# matrix kernel.  Results in gpurun_out/prof_rsnt_<tag>/ (copy what is to be kept into profiles/).   usage: tools/prof_rsnt.sh <tag>
set -u
TAG=${1:-r04}
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_rsnt_$TAG
mkdir -p "$OUT"
cd /tmp
python "$ROOT/tools/bench_rsnt.py" --only hann > "$OUT/bench_rsnt.json" 2> "$OUT/bench_rsnt.err"
rocprofv3 -f csv --kernel-trace --stats -d "$OUT/trace" -o trace -- python "$ROOT/tools/bench_rsnt.py" --only hann > "$OUT/trace.log" 2>&1
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > "$OUT/mfma_counters_available.txt"
rocprofv3 -f csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_mfma" -o mfma -- python "$ROOT/tools/bench_rsnt.py" --only hann > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 -f csv --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d "$OUT/pmc_valu" -o valu -- python "$ROOT/tools/bench_rsnt.py" --only hann > "$OUT/pmc_valu.log" 2>&1
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        lines.append(f"{r['Name'][:110]:110s} calls={int(r['Calls']):6d} avg_us={float(r['AverageNs']) / 1e3:9.1f} pct={100 * float(r['TotalDurationNs']) / tot:5.2f}")
open(os.path.join(out, "rsnt_kernel_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "reson" not in k: continue
        import re
        m = re.search(r"(\w+Kernel<\d+>)", k)
        a = acc[m.group(1) if m else k[:60]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
txt = ["RSNT render, Hann (3 vectors), cfg2 buffer: counters per launch (SQ_*_CYCLES of waves in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM in cycles)"]
for k, d in acc.items():
    txt.append("== " + k + "  " + "  ".join(f"{c} {v / n:.0f}" for c, (v, n) in sorted(d.items())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        busy, act = d["SQ_VALU_MFMA_BUSY_CYCLES"][0] / d["SQ_VALU_MFMA_BUSY_CYCLES"][1], d["GRBM_GUI_ACTIVE"][0] / d["GRBM_GUI_ACTIVE"][1]
        # (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: / 8 = the launch's duration in shader cycles)
        txt.append(f"   launch = {act / 8:.0f} cycles;  MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) = {busy / (act / 8 * 1024):.3f}")
open(os.path.join(out, "rsnt_mfma_counters.txt"), "w").write("\n".join(txt) + "\n")
print("\n".join(txt))
PY
