#!/bin/bash
# Where do the ~125 small device copies per training step come from?  HIP API + memory-copy trace of 3 steps
# (no counters in this run: traces and --pmc are never combined).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/copies
mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 2 --no-exclusive --no-infer --no-cpu-baseline > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/copies"
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    c = collections.Counter((r.get("Direction"), r.get("Size") or str(int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0)))) for r in rows)
    print(f, len(rows))
    for k, v in c.most_common(15):
        print("  ", k, v)
    print(rows[0].keys())
for f in glob.glob(out + "/**/*hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    c = collections.Counter(r["Function"] for r in rows)
    print(f)
    for k, v in c.most_common(25):
        print("  ", k, v)
PY
