#!/bin/bash
# configs[4] serving probe under rocprofv3: prints the per-batch time and the soft-NMS kernel's average duration.
# usage (GPU box, repo root): bash tools/probes/c4_nms_time.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}
OUT=$R/gpurun_out/nms/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/probes/c4_infer.py > $OUT/log.txt 2>&1
grep -h "ms_per_step" $OUT/log.txt | sed 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/'
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
grep -h "nms" $f | cut -d, -f1-4 | cut -c1-60,180-
