#!/bin/bash
# A/B of two builds of the library on ONE box: the probe build (librnet_hip_probe.so, built from the previous sources)
# against the current library, alternating processes.  Usage: tools/probes/ab_lib_step.sh [rounds]
R=${1:-2}
P=$PWD/retinanet-tensorflow2.x_amd/retinanet/librnet_hip_probe.so
for i in $(seq $R); do
  echo "== old (probe build)"; RNET_HIP_LIB=$P python tools/ab_step.py --variants auto --rounds 3 --steps 6 2>&1 | tail -2
  echo "== new"; python tools/ab_step.py --variants auto --rounds 3 --steps 6 2>&1 | tail -2
  echo "== old infer"; RNET_HIP_LIB=$P python tools/bench_infer.py 2>&1 | tail -1
  echo "== new infer"; python tools/bench_infer.py 2>&1 | tail -1
done
