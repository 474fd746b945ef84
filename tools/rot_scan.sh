#!/bin/bash
# K_A / step time against the number of distinct input buffers the launches rotate over (1: L2-resident; 2 .. 8: Infinity-Cache-resident; >= 12: HBM)
cd "$(dirname "$0")/.."
for n in 1 2 4 8 12 16 24; do echo -n "buffers=$n: "; SGZ_BUFFERS=$n python tools/ka_time.py 60 2>/dev/null | tail -1 | python -c "
import ast,sys
d=ast.literal_eval(sys.stdin.read())
print(' '.join(f\"{k} {v['ka_us']:.2f} step {v['step_us']:.2f}\" for k,v in d.items() if k=='cfg2_348'))"; done
