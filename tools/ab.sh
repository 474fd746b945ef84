#!/bin/bash
# A/B timing of two builds of the library on ONE box (boxes differ by ~1 us at cfg2): tools/ab.sh <libA.so> <libB.so> [rounds]
# alternates the two, prints K_A at cfg2 / tail-free / cfg5 chunk for each round.  Build variants with
#   python signalizer_amd/build.py --force && cp signalizer_amd/libsgz.so signalizer_amd/build/libsgz_A.so
A=$1; B=$2; N=${3:-3}
cd "$(dirname "$0")/.."
for r in $(seq 1 $N); do
  for L in "$A" "$B"; do
    echo -n "$(basename $L): "
    SGZ_LIB=$(pwd)/$L timeout 200 python tools/ka_time.py 60 2>&1 | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print(' '.join(f\"{k} {v['ka_us']:.2f}/{v['ka_min_us']:.2f} step {v['step_us']:.2f}\" for k,v in d.items()))"
  done
done
