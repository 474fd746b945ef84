#!/bin/bash
# VGPR / spill / occupancy summary of every kernel of one HIP source (default spectrum_fft.hip), from hipcc's resource-usage remarks
src=${1:-$(dirname "$0")/../signalizer_amd/csrc/spectrum_fft.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
    -Rpass-analysis=kernel-resource-usage -x hip -c "$src" -o /tmp/_ru.o 2>&1 |
  sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/{name=$NF} / VGPRs: /{v=$NF} /Occupancy/{occ=$NF} /SGPRs Spill:/{ss=$NF} /VGPRs Spill:/{sp=$NF} /LDS Size/{printf "%-64s VGPRs %3s  waves/SIMD %s  vgpr-spill %s  sgpr-spill %s\n", name, v, occ, sp, ss}' |
  while read -r name rest; do printf "%-78s %s\n" "$(echo "$name" | c++filt | cut -c1-78)" "$rest"; done
