# exclusive (one-stream) time of the weight-gradient kernel families with two builds of the library on one box
for r in 1 2; do for L in "" "$1"; do echo -n "lib=${L:-default}  "; RNET_HIP_LIB=$L python bench.py --no-infer --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d['roofline']['wgrad']['exclusive']; print(d['ms_per_step'], w['ms_per_step'], {k[:28]:v['ms_per_step'] for k,v in w['kernels'].items()})"; done; done
