# per-layer: K step 64 (two or three stages) vs 32 (four stages) on the 128-row kernel
for p in g2_a g3_a g4_a g4_3x3 g3_3x3 g2b0_a g1_a; do for b in 8 32; do for v in "RNET_X=0" "RNET_CONV128_BK=32"; do echo -n "$v: "; env $v python tools/bench_conv.py --preset $p --batch $b --iters 50 --tile 1 2>&1 | tail -1; done; done; done
