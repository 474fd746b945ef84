#!/bin/bash
TAG=${1:-x}
cd "$(dirname "$0")/.."
ROOT=$(pwd); export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmcmem_$TAG; mkdir -p "$OUT"; cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 -f csv --pmc $set --kernel-trace -d "$OUT/$name" -o c -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/$name.log" 2>&1
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
