#!/bin/bash
# A/B of the weight-gradient stream's CU caps (TrainEngine: RNET_WGRAD_CUS="wide kernels,128-tile kernel"), same box, two runs each
mkdir -p gpurun_out/r6i
for cus in ${CUS_LIST:-"176,256" "160,240" "144,224" "128,208" "128,192" "112,192" "144,192" "160,208" "144,256" "96,160"}; do
  for rep in 1 2; do
    echo -n "RNET_WGRAD_CUS=$cus : "
    RNET_WGRAD_CUS=$cus python bench.py --steps 20 --warmup 5 --no-infer --no-extras --no-cpu-baseline --no-probe --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
