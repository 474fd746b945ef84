#!/bin/bash
# Ablation builds of the channel-split K_A (results are WRONG by construction: timing only): tools/ab/libsgz_abl<N>.so for
#   1 no pass-3 butterflies  2 no pass-2 twiddles  3 no window  4 no recombination arithmetic  5 no pixel map  6 no exchange 2
#   7 no pass-1 butterflies  8 no pass-2 butterflies  9 no exchange 1
# then on the GPU box:  for n in 0 1 ...; do SGZ_LIB=$PWD/tools/ab/libsgz_abl$n.so python tools/ka_time.py 40; done
cd "$(dirname "$0")/.."
mkdir -p tools/ab
for n in "$@"; do
  SGZ_EXTRA_HIPCC_FLAGS="-DSGZ_ABL=$n" python signalizer_amd/build.py --force > /dev/null 2>&1 && cp signalizer_amd/libsgz.so tools/ab/libsgz_abl$n.so && echo built $n
done
python signalizer_amd/build.py --force > /dev/null 2>&1
