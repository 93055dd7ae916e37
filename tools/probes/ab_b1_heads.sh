# head / FPN-output launches at small batches: the dispatcher's choice without and with a split-K workspace attached
for b in 1 2 4; do
for p in tower pred_class fpn_out; do
  for s in "" "--splitk"; do
   echo -n "B=$b $p $s: "; python tools/bench_conv.py --preset $p --batch $b --iters 50 $s --bias 2>&1 | tail -1
  done
done
done
