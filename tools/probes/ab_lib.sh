# same-box A/B of two builds of librnet_hip.so on the training bench: bash tools/probes/ab_lib.sh <other .so>
for r in 1 2; do for L in "" "$1"; do echo -n "lib=${L:-default}  "; RNET_HIP_LIB=$L python bench.py --no-infer --no-cpu-baseline --no-exclusive 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
