#!/bin/bash
# LDS bank-conflict counters per kernel over two training steps (one-stream backward); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_lds_step; mkdir -p $OUT; cd $R
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds -- python3 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline > $OUT/lds.log 2>&1
find $OUT -name "*counter_collection.csv" | wc -l
