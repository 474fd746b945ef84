#!/bin/bash
# like ab.sh for any number of builds: tools/ab3.sh <rounds> <lib.so>[:narrow] [<lib.so>[:narrow] ...]
# (":narrow" = plan option SGZ_OPT_WIDE_GROUPS 0: the 512-thread form of the N = 32768 channel-split kernel)
N=$1; shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $N); do
  for A in "$@"; do
    L=${A%%:*}; NARROW=0; [ "$A" != "$L" ] && NARROW=1
    echo -n "$(basename $A): "
    SGZ_NARROW=$NARROW SGZ_LIB=$(pwd)/$L timeout 200 python tools/ka_time.py 60 2>&1 | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print(' '.join(f\"{k} {v['ka_us']:.2f}/{v['ka_min_us']:.2f} step {v['step_us']:.2f}\" for k,v in d.items()))"
  done
done
