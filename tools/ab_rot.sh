#!/bin/bash
# tools/ab_rot.sh <rounds> <buffers> <lib.so> ...: K_A at cfg2 with the launches rotating over <buffers> input copies
N=$1; B=$2; shift; shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $N); do for L in "$@"; do echo -n "$(basename $L) buffers=$B: "; SGZ_BUFFERS=$B SGZ_LIB=$(pwd)/$L python tools/ka_time.py 60 2>/dev/null | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print(' '.join(f\"{k} {v['ka_us']:.2f} step {v['step_us']:.2f}\" for k,v in d.items() if k=='cfg2_348'))"; done; done
