#!/bin/bash
# extra PMC passes for one conv preset (tools/bench_conv.py), run on the GPU box from the repo root:
#   bash tools/probes/pmc_extra.sh [preset ...]      (default: tower)
# instruction mix, LDS bank conflicts, wait cycles — one counter group per pass, never combined with traces
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for PRESET in ${@:-tower}; do
  OUT=$R/gpurun_out/pmc_extra/$PRESET; mkdir -p $OUT
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/insts -- python3 tools/bench_conv.py --preset $PRESET --batch 32 --iters 3 > $OUT/insts.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -- python3 tools/bench_conv.py --preset $PRESET --batch 32 --iters 3 > $OUT/lds.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/wait -- python3 tools/bench_conv.py --preset $PRESET --batch 32 --iters 3 > $OUT/wait.log 2>&1
done
find $R/gpurun_out/pmc_extra -name "*counter_collection.csv" | wc -l
