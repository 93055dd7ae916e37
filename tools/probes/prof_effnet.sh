# EfficientNet-B3 640^2 (BASELINE config 5, mixed_float16 on librnet_hip_f16.so): kernel-trace stats of tools/bench_effnet.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_effb3 -- python3 tools/bench_effnet.py --iters 5 > gpurun_out/prof_effb3.log 2>&1
grep -E "train step|inference" gpurun_out/prof_effb3.log | tail -3
