#!/bin/bash
# The -DSGZ_DEBUG build of the library (phase clocks, per-workgroup schedule stamps, sgz_debug_* entry points) -> tools/ab/lib_dbg.so:
# spectrum_real.hip and api.hip recompiled with the flag, linked with the current objects of the other translation units.
# usage: tools/mkdebug.sh [extra hipcc flags ...]
set -e
cd "$(dirname "$0")/.."
B=signalizer_amd/build
python signalizer_amd/build.py > /dev/null
mkdir -p tools/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -x hip -DSGZ_DEBUG"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c signalizer_amd/csrc/spectrum_real.hip -o /tmp/dbg_real.o "$@" &
/opt/rocm/bin/hipcc $F -c signalizer_amd/csrc/api.hip -o /tmp/dbg_api.o "$@" &
wait
OBJS=$(ls $B/*.o | grep -v "/spectrum_real.hip.o" | grep -v "/api.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib_dbg.so $OBJS /tmp/dbg_real.o /tmp/dbg_api.o -ldl
echo tools/ab/lib_dbg.so
