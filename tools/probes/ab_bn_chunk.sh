#!/bin/bash
# elementwise BatchNorm passes: grid-stride (U0) against one workgroup per 256 * U contiguous units (probe builds with
# -DRN_BN_CHUNK_U=<U>; the library's own build is U = 4)
D=$PWD/retinanet-tensorflow2.x_amd/retinanet
for i in 1 2; do
  for v in _u0 "" _u8 _u16 _u32; do
    L=$D/librnet_hip_probe$v.so
    echo -n "${v:-_u4}  "; RNET_HIP_LIB=$L python tools/ab_step.py --variants auto --rounds 3 --steps 6 2>&1 | tail -1
  done
done
