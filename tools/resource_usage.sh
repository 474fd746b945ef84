#!/bin/bash
# VGPR / spill summary of every kernel of one HIP source (default spectrum_fft.hip), as hipcc's resource-usage remarks report it
src=${1:-/root/repo/signalizer_amd/csrc/spectrum_fft.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
    -Rpass-analysis=kernel-resource-usage -x hip -c "$src" -o /tmp/_ru.o 2>&1 |
  awk '/Function Name:/{name=$NF} / VGPRs:/{v=$(NF-1)} /VGPRs Spill:/{sp=$(NF-1)} /SGPRs Spill:/{ss=$(NF-1)} /Occupancy/{printf "%-70s VGPRs %s  vspill %s  sspill %s\n", name, v, sp, ss}'
