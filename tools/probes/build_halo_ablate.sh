#!/bin/bash
# Probe builds of librnet_hip.so with parts of conv_halo_kernel's inner loop removed (timing only, wrong results):
#   tools/probes/bin/librnet_halo_ab<N>.so, N = HALO_ABLATE bitmask (see rn_conv_halo.hip).
# Only rn_conv_halo.hip is recompiled; the other objects come from the normal build (run `make` first).
set -e
cd "$(dirname "$0")/../../retinanet-tensorflow2.x_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DHALO_ABLATE=$n -DHALO_PROF -c rn_conv_halo.hip -o /tmp/rn_conv_halo_ab$n.o
  objs=$(ls build/*.o | grep -v rn_conv_halo.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/probes/bin/librnet_halo_ab$n.so $objs /tmp/rn_conv_halo_ab$n.o
done
