#!/bin/bash
# SQ/LDS counters for the bench kernels: tools/pmc.sh <tag>
TAG=${1:-x}
EXTRA=${2:-}          # e.g. "--workload cfg5"
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 -f csv --pmc $set --kernel-trace -d "$OUT/$name" -o c -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras $EXTRA > "$OUT/$name.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sgz" not in k: continue
        a = acc[k.split("(")[0][-40:]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print("==", k)
    for c, (v, n) in sorted(d.items()):
        print(f"   {c:28s} avg/dispatch = {v / n:16.1f}")
PY
