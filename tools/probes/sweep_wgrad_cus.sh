#!/bin/bash
# RNET_WGRAD_CUS sweep on one box (separate processes, two passes): "<wide kernels' workgroups>,<128-tile kernel's>"
for i in 1 2; do
  for v in "128,256" "144,256" "160,256" "176,256" "192,256" "208,256" "176,128" "176,192" "176,384" "256,1024"; do
    echo -n "RNET_WGRAD_CUS=$v  "; RNET_WGRAD_CUS=$v python tools/ab_step.py --variants auto --rounds 3 --steps 6 2>&1 | tail -1
  done
done
