#!/bin/bash
# run on the GPU box: K_A time of every ablation build present (tools/ablate_builds.sh)
cd "$(dirname "$0")/.."
for n in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  L=tools/ab/libsgz_abl$n.so
  [ -f $L ] || continue
  echo -n "abl $n: "
  SGZ_LIB=$PWD/$L timeout 200 python tools/ka_time.py 40 2>&1 | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print(' '.join(f\"{k} {v['ka_us']:.2f}/{v['ka_min_us']:.2f}\" for k,v in d.items()))"
done
