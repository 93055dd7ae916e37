#!/bin/bash
# step time with two builds of the library, alternating processes on one box: tools/probes/ab_two_libs.sh <libA.so> <libB.so>
for i in 1 2 3; do
  for L in "$1" "$2"; do
    echo -n "$(basename $L)  "; RNET_HIP_LIB=$PWD/$L python tools/ab_step.py --variants auto --rounds 3 --steps 6 2>&1 | tail -1
  done
done
