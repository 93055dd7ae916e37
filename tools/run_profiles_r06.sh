#!/bin/bash
# Round-6 profile recipe (one gpurun call, run from the repo root on the GPU box): the bench line, kernel-trace stats of the
# training / inference legs and the PMC passes of tools/run_profiles.sh, kernel-trace stats + one FETCH_SIZE pass for
# BASELINE configs[3] / configs[4], the per-launch tables of the inference engine (now with the fused stage-1 blocks), the
# per-queue gap analysis of a two-stream training trace (tools/trace_gaps.py), the stage-1 block kernel's own timings.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
bash tools/run_profiles.sh > $OUT/run_profiles.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $R/bench.py --size 1024 --train-batch 16 --steps 5 --warmup 2 --no-exclusive --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/c3.log 2>&1
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c3_pmc_fetch -- python3 $R/bench.py --size 1024 --train-batch 16 --steps 2 --warmup 1 --no-exclusive --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/c3_pmc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $R/tools/bench_effnet.py --iters 5 > $OUT/c4.log 2>&1
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4_pmc_fetch -- python3 $R/tools/bench_effnet.py --iters 2 > $OUT/c4_pmc.log 2>&1
# two-stream training trace (no stats: the raw kernel trace) -> per-queue busy / idle
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 6 --warmup 3 --no-exclusive --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/trace.log 2>&1
cd $R
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $f --steps 2 > $OUT/trace_gaps.json 2> $OUT/trace_gaps.err
rm -rf $OUT/trace
find $OUT -name "*kernel_stats.csv" | head -20
python3 tools/profile_layers.py --batch 1 --iters 20 > $OUT/layers_b1.txt 2>&1
python3 tools/profile_layers.py --batch 8 --iters 10 > $OUT/layers_b8.txt 2>&1
python3 tools/bench_bneck.py > $OUT/bneck_bench.txt 2>&1
python3 tools/bench_dw.py --forms wgrad > $OUT/dw_wgrad.txt 2>&1
python3 tools/bench_dw.py --forms fwd,dgrad > $OUT/dw_fwd_dgrad.txt 2>&1
python3 tools/bench_bn.py > $OUT/bn_bench.txt 2>&1
python3 tools/bench_effnet.py --iters 6 > $OUT/effnet_bench.txt 2>&1
bash tools/probes/c4_nms_time.sh final > $OUT/c4_nms_time.txt 2>&1
tail -c 400 $OUT/bench_line.json
