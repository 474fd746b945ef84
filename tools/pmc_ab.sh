#!/bin/bash
# counters + timing of K_A for several builds on one box: tools/pmc_ab.sh <tag> <lib.so> [<lib.so> ...]   (SGZ_WIDE=1: the 1024-thread form)
TAG=$1; shift
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
for L in "$@"; do
  name=$(basename $L .so)
  OUT=$ROOT/gpurun_out/pmcab_$TAG/$name
  mkdir -p "$OUT"
  export SGZ_LIB=$ROOT/$L
  ( cd /tmp
    rocprofv3 -f csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d "$OUT/a" -o c -- python "$ROOT/tools/ka_pmc.py" cfg2 10 > "$OUT/a.log" 2>&1
    rocprofv3 -f csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/b" -o c -- python "$ROOT/tools/ka_pmc.py" cfg2 10 > "$OUT/b.log" 2>&1 )
  echo "== $name"
  python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "stftReal" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("   " + "  ".join(f"{c.replace('SQ_', '')} {v / n / 696:.0f}" for c, (v, n) in sorted(acc.items())) + "   (per workgroup)")
PY
  timeout 300 python tools/ka_time.py 60 2>&1 | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print('   ' + ' '.join(f\"{k} {v['ka_us']:.2f}/{v['ka_min_us']:.2f} step {v['step_us']:.2f}\" for k,v in d.items()))"
done
