#!/bin/bash
# A/B two builds of libsgz.so on the GPU box: tools/ab.sh <other.so>
cd "$(dirname "$0")/.."
python tools/ablate.py 2>&1 | grep -E "^full|no map" 
cp signalizer_amd/libsgz.so /tmp/a.so; cp "$1" signalizer_amd/libsgz.so
echo "--- $1"
python tools/ablate.py 2>&1 | grep -E "^full|no map"
cp /tmp/a.so signalizer_amd/libsgz.so
