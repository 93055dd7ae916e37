#!/bin/bash
# EfficientNet-B3 training step with two builds of the f16 library, alternating processes on one box:
# bash tools/probes/ab_effnet_libs.sh <libA_f16.so> <libB_f16.so>
for i in 1 2; do
  for L in "$1" "$2"; do
    echo -n "$(basename $L)  "; RNET_HIP_LIB_F16=$PWD/$L python tools/bench_effnet.py --iters 6 2>&1 | grep -E "train step|back to back" | tail -2 | tr '\n' ' '; echo
  done
done
