#!/bin/bash
# long random-sweep campaign: failures only -> gpurun_out/fuzz_campaign.txt     usage: fuzz_campaign.sh <first seed> <seeds> 
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/fuzz_campaign.txt
: > $out
first=${1:-100}; n=${2:-10}
for ((s=first; s<first+n; s++)); do
  for cmd in "fuzz_parity.py 300 $s" "fuzz_parity.py 300 $s wild" "fuzz_stages.py 150 $s" "fuzz_realtime.py 100 $s" "fuzz_scope.py 100 $s" "fuzz_rsnt.py 60 $s"; do
    echo "== $cmd" >> $out
    timeout 900 python tools/$cmd 2>&1 | grep -v " ok " | grep -v amdgpu.ids | cut -c1-600 >> $out
  done
done
grep -c BAD $out
tail -3 $out
