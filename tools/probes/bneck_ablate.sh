#!/bin/bash
# which part of rn_bneck.hip's row loop costs the time: the probe build's ablated kernels (make -C csrc probe)
export RNET_HIP_LIB=$GRAFT_REPO_ROOT/retinanet-tensorflow2.x_amd/retinanet/librnet_hip_probe.so
for ab in 0 1 2 3 4 7 8 16 32 64 72 127; do
  echo "--- ablate=$ab"; python tools/bench_bneck.py --ablate $ab --batches 32 2>&1 | grep "Cx="
done
