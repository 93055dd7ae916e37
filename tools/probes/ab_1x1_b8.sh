# 1x1 launches that the dispatcher sends to conv_big_kernel (>= 192 tiles of 256 rows) at serving batches vs the 128-row kernel
for p in g1_out g1_sc g2_out g3_out g3_sc fpn_lat; do for b in 8 32; do for m in 0 1000000; do echo -n "$p B=$b min_tiles=$m: "; python tools/bench_conv.py --preset $p --batch $b --iters 50 --splitk --min-tiles $m 2>&1 | tail -1; done; done; done
