# A/B of the stage-2 BatchNorm reduction inside the training step (one-stream kernel trace): RNET_BN_FINAL=1 never splits
mkdir -p gpurun_out/r05b; cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
export RNET_BN_FINAL=$f; export RNET_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05b/fin$f -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-exclusive --no-infer --no-cpu-baseline --no-probe --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r05b/fin$f.log 2>&1
grep "bn_colreduce_final" $GRAFT_REPO_ROOT/gpurun_out/r05b/fin$f/runc_kernel_stats.csv | cut -c1-120
done
unset RNET_WGRAD_STREAM
cd $GRAFT_REPO_ROOT
python tools/ab_engines.py --a RNET_BN_FINAL=1 --b RNET_BN_FINAL=0 --steps 10 --rounds 4 2>&1 | tail -8
