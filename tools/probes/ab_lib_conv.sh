#!/bin/bash
# old (probe build) vs new library on single conv launches.  Usage: tools/probes/ab_lib_conv.sh "<bench_conv args>"
P=$PWD/retinanet-tensorflow2.x_amd/retinanet/librnet_hip_probe.so
for i in 1 2; do
  echo "== old: $1"; RNET_HIP_LIB=$P python tools/bench_conv.py $1 2>&1 | grep -v amdgpu.ids
  echo "== new: $1"; python tools/bench_conv.py $1 2>&1 | grep -v amdgpu.ids
done
