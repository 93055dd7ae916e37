#!/bin/bash
# Round-end check (CPU, hipcc only): every kernel's VGPRs / scratch / occupancy from -Rpass-analysis=kernel-resource-usage,
# for both builds of the library.  Prints the kernels with scratch memory (spills or run-time indexed private arrays:
# the depthwise weight gradient lived in 304-416 bytes of scratch per lane for two rounds) and the ones at one wave per
# SIMD outside the persistent GEMM kernels (the 5x5 depthwise kernel, before its filter-row loop was rolled).
#   bash tools/scan_kernel_resources.sh
cd "$(dirname "$0")/../retinanet-tensorflow2.x_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
bad=0
for def in "" "-DRN_F16"; do
  for f in rn_*.hip; do
    /opt/rocm/bin/hipcc $F $def -c $f -o /tmp/scan_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
      grep -E "Function Name:| VGPRs:|ScratchSize|Occupancy" | sed 's/remark: [^ ]* //g; s/\[-Rpass[^]]*\]//g' | paste - - - - |
      awk -v f="$f" -v d="${def:-bf16}" '{ name=$3; vg=$6; sc=$10; occ=$NF;
           if (sc+0 > 0) { print d, f, name, "VGPRs", vg, "scratch", sc, "occupancy", occ; bad=1 }
           else if (occ+0 <= 1 && name !~ /conv_(big|halo)_kernel|wgrad_big/) print d, f, name, "VGPRs", vg, "occupancy 1" }'
  done
done
rm -f /tmp/scan_$$.o
