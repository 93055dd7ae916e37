#!/bin/bash
# bench.py with and without environment switches, alternating processes on ONE box:
#   tools/probes/ab_env_bench.sh OUTDIR "RNET_PRED_PAIR=0 RNET_GROUP_ORDER=graph" [extra bench.py arguments]
OUT=${1:-gpurun_out/abenv}; SW=$2; shift 2
mkdir -p $OUT
for i in 1 2; do
  env $SW python bench.py --no-cpu-baseline --no-probe --no-extras "$@" > $OUT/bench_base_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-probe --no-extras "$@" > $OUT/bench_new_$i.json 2>/dev/null
done
