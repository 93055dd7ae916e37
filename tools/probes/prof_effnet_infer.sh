cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_eff_inf -- python3 tools/bench_effnet.py --iters 0 > gpurun_out/prof_eff_inf.log 2>&1
tail -2 gpurun_out/prof_eff_inf.log
