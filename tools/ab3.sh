#!/bin/bash
# like ab.sh for any number of builds: tools/ab3.sh <rounds> <lib.so> [<lib.so> ...]
N=$1; shift
cd "$(dirname "$0")/.."
for r in $(seq 1 $N); do
  for L in "$@"; do
    echo -n "$(basename $L): "
    SGZ_LIB=$(pwd)/$L timeout 200 python tools/ka_time.py 60 2>&1 | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print(' '.join(f\"{k} {v['ka_us']:.2f}/{v['ka_min_us']:.2f} step {v['step_us']:.2f}\" for k,v in d.items()))"
  done
done
