#!/bin/bash
# same-box A/B of engine switches (training leg of bench.py, ms per step, three runs each, interleaved)
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-infer --no-extras --no-cpu-baseline --no-probe --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"; }
for rep in 1 2 3; do
  echo -n "default            : "; run RNET_NOOP=1
  echo -n "RNET_FUSE_BN_BWD=2 : "; run RNET_FUSE_BN_BWD=2
  echo -n "RNET_SPLITK=1      : "; run RNET_SPLITK=1
  echo -n "RNET_WGRAD_GROUP=0 : "; run RNET_WGRAD_GROUP=0
  echo -n "RNET_WGRAD_GROUP=all : "; run RNET_WGRAD_GROUP=all
done
