# the 128-row kernel's K loop (same box, separate processes): RNET_CONV128_STAGES=2 = two LDS stages with a full wait per
# step (rounds 1 - 4); default = as many stages as leave two workgroups per CU, counted waits, K step 32 on the shallow layers;
# RNET_CONV128_BK=64 = the new loop with K step 64 wherever the channel count allows
for r in 1 2; do for v in "RNET_CONV128_STAGES=2" "RNET_CONV128_BK=64" "RNET_X=0"; do for b in 1 8; do echo -n "$v B=$b: "; env $v python tools/bench_infer.py --batch $b 2>&1 | tail -1; done; done; done
for r in 1 2; do for v in "RNET_CONV128_STAGES=2" "RNET_CONV128_BK=64" "RNET_X=0"; do echo -n "$v train: "; env $v python bench.py --steps 20 --warmup 5 --no-infer --no-extras --no-cpu-baseline --no-exclusive 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | head -1; done; done
