#!/bin/bash
# round-4 A/B on ONE box, alternating processes: a baseline build of the library (tools/probes/bin/librnet_base.so: the
# previous sources, same ABI) against the current one.  Usage: tools/probes/ab_r04.sh OUTDIR [extra bench.py arguments]
OUT=${1:-gpurun_out/r04ab}; shift
mkdir -p $OUT
P=$PWD/tools/probes/bin/librnet_base.so
for i in 1 2; do
  RNET_HIP_LIB=$P python bench.py --no-cpu-baseline --no-probe --no-extras "$@" > $OUT/bench_base_$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --no-probe --no-extras "$@" > $OUT/bench_new_$i.json 2>/dev/null
done
