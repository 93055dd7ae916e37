#!/bin/bash
# bench.py on another build of the library (same ABI) against the current one, alternating processes on ONE box:
#   tools/probes/ab_lib_bench.sh OUTDIR path/to/other.so [extra bench.py arguments]
OUT=${1:-gpurun_out/ablib}; P=$(realpath $2); shift 2
mkdir -p $OUT
for i in 1 2; do
  RNET_HIP_LIB=$P python bench.py --no-cpu-baseline --no-probe --no-extras "$@" > $OUT/bench_other_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-probe --no-extras "$@" > $OUT/bench_new_$i.json 2>/dev/null
done
