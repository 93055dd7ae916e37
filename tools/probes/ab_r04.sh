#!/bin/bash
# round-4 A/B on ONE box, alternating processes: (a) the halo kernel before the shared set-up tables (tools/probes/bin/
# librnet_prep3.so, whole tiles), (b) the current library with whole tiles, (c) the current library with the split last round
OUT=${1:-gpurun_out/r04f}
mkdir -p $OUT
P=$PWD/tools/probes/bin/librnet_prep3.so
for i in 1 2; do
  RNET_SPLITK=0 RNET_HIP_LIB=$P python bench.py --no-cpu-baseline --no-probe > $OUT/bench_prep3_$i.json 2>/dev/null
  RNET_SPLITK=0 python bench.py --no-cpu-baseline --no-probe > $OUT/bench_whole_$i.json 2>/dev/null
  RNET_SPLITK=1 python bench.py --no-cpu-baseline --no-probe > $OUT/bench_split_$i.json 2>/dev/null
done
