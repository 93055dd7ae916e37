#!/bin/bash
# Round profiles (run on the GPU box from the repo root): kernel-trace stats of the bench command, then
# FETCH_SIZE / WRITE_SIZE PMC passes (one counter group per pass, never combined with traces; one-stream backward).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
# the bench's own training leg (default --steps 20 --warmup 3), without the extra one-stream step: every step in the
# trace is a two-stream step like the timed ones, so the dominant kernel's average here is the bench line's avg_launch_us
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 bench.py --no-exclusive --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/train.log 2>&1
# the same command with the one-stream backward: kernel durations without the wgrad launches of the second stream
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train1s -- python3 bench.py --steps 5 --warmup 2 --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/train1s.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infer -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-probe --no-extras > $OUT/infer.log 2>&1
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/pmc_fetch.log 2>&1
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/pmc_write.log 2>&1
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/pmc_mfma.log 2>&1
# what the matrix pipe sustains on this box: MFMA-only probe, random vs zero operands, with the core clock it ran at
python3 tools/probes/mfma_sustained.py > $OUT/mfma_probe.json 2> $OUT/mfma_probe.err
find $OUT -name "*.csv" | head -30
tail -2 $OUT/train.log
