# single-segment 3x3 launches of ResNet stage 3 / 4 at serving batches, split-K workspace attached: the dispatcher's choice
# (halo kernel, tiles split) vs the 128-row kernel (conv_big_min_tiles = 1e6 keeps it there; its own split-K allowed)
for p in g3_3x3 g4_3x3; do for b in 2 4 8 16; do for m in 0 1000000; do echo -n "$p B=$b min_tiles=$m: "; python tools/bench_conv.py --preset $p --batch $b --iters 50 --splitk --min-tiles $m 2>&1 | tail -1; done; done; done
