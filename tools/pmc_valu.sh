#!/bin/bash
# VALU instruction count / busy time per kernel of any command: tools/pmc_valu.sh <tag> -- <command ...>
TAG=$1; shift; shift
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcv_$TAG
mkdir -p "$OUT"
( cd /tmp && rocprofv3 -f csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT" -o c -- "$@" > "$OUT/log.txt" 2>&1 )
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sgz" not in k: continue
        a = acc[k.split("(")[0][-44:]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print("==", k, "  ".join(f"{c} {v / n:.0f}" for c, (v, n) in sorted(d.items())))
PY
