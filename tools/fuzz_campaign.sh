#!/bin/bash
# long random-sweep campaign: failures only -> gpurun_out/fuzz_campaign.txt     usage: [LOAD=n LOADSECS=1500] fuzz_campaign.sh <first seed> <seeds>
# LOAD=n: n other PROCESSES render flat out on the same device for the whole campaign (tools/gpu_load.py): every case runs time-sliced
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/fuzz_campaign.txt
: > $out
first=${1:-100}; n=${2:-10}
pids=()
for ((k=0; k<${LOAD:-0}; k++)); do
  kind=spectrum; [ $((k % 2)) = 1 ] && kind=rsnt
  python tools/gpu_load.py ${LOADSECS:-1500} $kind > /dev/null 2>&1 &
  pids+=($!)
done
[ ${#pids[@]} -gt 0 ] && sleep 20 && echo "== ${#pids[@]} load processes" >> $out
for ((s=first; s<first+n; s++)); do
  for cmd in "fuzz_parity.py 300 $s" "fuzz_parity.py 300 $s wild" "fuzz_stages.py 150 $s" "fuzz_realtime.py 100 $s" "fuzz_scope.py 100 $s" "fuzz_rsnt.py 60 $s"; do
    echo "== $cmd" >> $out
    timeout 900 python tools/$cmd 2>&1 | grep -v " ok " | grep -v amdgpu.ids | cut -c1-600 >> $out
  done
done
for p in "${pids[@]}"; do kill $p; done
grep -c BAD $out
tail -3 $out
