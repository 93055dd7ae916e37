#!/bin/bash
# Round-5 profile recipe (one gpurun call, run from the repo root on the GPU box): the bench line, kernel-trace stats of the
# training / inference legs, the PMC passes of tools/run_profiles.sh, and — new this round — kernel-trace stats + one PMC pass
# for BASELINE configs[3] (ResNet50-1024^2) and configs[4] (EfficientNet-B3 f16), so that extra.config3 / extra.config4 of
# the bench line can be recomputed from profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
bash tools/run_profiles.sh > $OUT/run_profiles.log 2>&1
cd /tmp
# configs[3]: 1024^2, 16 images — two-stream trace (as timed) and FETCH_SIZE pass
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $R/bench.py --size 1024 --train-batch 16 --steps 5 --warmup 2 --no-exclusive --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/c3.log 2>&1
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c3_pmc_fetch -- python3 $R/bench.py --size 1024 --train-batch 16 --steps 2 --warmup 1 --no-exclusive --no-infer --no-cpu-baseline --no-probe --no-extras > $OUT/c3_pmc.log 2>&1
# configs[4]: EfficientNet-B3 640^2 f16 — two-stream trace (7 steps + 23 serving replays) and FETCH_SIZE pass
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $R/tools/bench_effnet.py --iters 5 > $OUT/c4.log 2>&1
RNET_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4_pmc_fetch -- python3 $R/tools/bench_effnet.py --iters 2 > $OUT/c4_pmc.log 2>&1
cd $R
find $OUT -name "*kernel_stats.csv" | head -20
# batch-1 / batch-8 per-launch tables of the inference engine
python3 tools/profile_layers.py --batch 1 --iters 20 > $OUT/layers_b1.txt 2>&1
python3 tools/profile_layers.py --batch 8 --iters 10 > $OUT/layers_b8.txt 2>&1
tail -c 400 $OUT/bench_line.json
