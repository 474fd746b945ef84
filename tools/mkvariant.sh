#!/bin/bash
# A variant build of the library for A/B work on one box: tools/mkvariant.sh <name> <file.hip> [extra hipcc flags ...]
# recompiles ONE translation unit with the flags and links it with the current objects of the others -> tools/ab/lib_<name>.so
set -e
NAME=$1; SRC=$2; shift; shift
cd "$(dirname "$0")/.."
B=signalizer_amd/build
python signalizer_amd/build.py > /dev/null
EXTRA=""
case "$SRC" in spectrum_fft.hip|spectrum_real.hip|spectrum_real16.hip|resonator.hip) EXTRA="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt $EXTRA -x hip -c signalizer_amd/csrc/$SRC -o /tmp/variant_$NAME.o "$@"
OBJS=$(ls $B/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/lib_$NAME.so $OBJS /tmp/variant_$NAME.o -ldl
echo tools/ab/lib_$NAME.so
