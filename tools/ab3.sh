#!/bin/bash
# like ab.sh for any number of builds: tools/ab3.sh <rounds> <lib.so>[:wide] [<lib.so>[:wide] ...]
# (":wide" = plan option SGZ_OPT_WIDE_GROUPS 1: the 1024-thread form of the N = 32768 channel-split kernel)
N=$1; shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $N); do
  for A in "$@"; do
    L=${A%%:*}; WIDE=0; [ "$A" != "$L" ] && WIDE=1
    echo -n "$(basename $A): "
    SGZ_WIDE=$WIDE SGZ_LIB=$(pwd)/$L timeout 200 python tools/ka_time.py 60 2>&1 | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print(' '.join(f\"{k} {v['ka_us']:.2f}/{v['ka_min_us']:.2f} step {v['step_us']:.2f}\" for k,v in d.items()))"
  done
done
