#!/bin/bash
# step time with and without one environment switch, alternating processes on one box:
#   tools/probes/ab_env.sh RNET_WGRAD_SIDE=halo [rounds]
for i in $(seq ${2:-3}); do
  echo -n "default      "; python tools/ab_step.py --variants auto --rounds 3 --steps 6 2>&1 | tail -1
  echo -n "$1  "; env "$1" python tools/ab_step.py --variants auto --rounds 3 --steps 6 2>&1 | tail -1
done
