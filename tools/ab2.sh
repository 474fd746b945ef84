#!/bin/bash
cd "$(dirname "$0")/.."
for r in 1 2 3; do
python tools/ablate.py 2>&1 | grep -E "^full  "
cp signalizer_amd/libsgz.so /tmp/a.so; cp "$1" signalizer_amd/libsgz.so
python tools/ablate.py 2>&1 | grep -E "^full  " | sed 's/full/other/'
cp /tmp/a.so signalizer_amd/libsgz.so
done
