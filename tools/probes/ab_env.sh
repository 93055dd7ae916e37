# same-box A/B of an environment switch on the training bench: bash tools/probes/ab_env.sh RNET_FUSE_BN_BWD
V=$1
for r in 1 2; do for x in 0 1; do echo -n "$V=$x  "; env $V=$x python bench.py --no-infer --no-cpu-baseline --no-exclusive 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; done; done
