"""Within-process A/B of the whole training step (run on the GPU box): the boxes of the pool differ by several percent,
so two bench.py runs cannot rank two variants.  ONE engine (BASELINE configs[2] shard, B = 32 at 640 x 640); a variant
is a set of rn_launch_opts fields written into the engine's live weight-gradient / convolution problem descriptors
between rounds; variants are timed in interleaved rounds and the median ms / step reported.

    RNET_AB_WORKSPACES=1 python tools/ab_step.py --variants "auto;wgrad_kernel=3;wgrad_kernel=1" [--rounds 5] [--steps 6]

`field=value` pairs separated by commas apply to the weight-gradient problems; prefix `conv.` for the forward / dgrad
problems (e.g. conv.conv_no_halo=1), `halo.` / `big.` / `small.` for the weight-gradient problems on that kernel only.  With RNET_HIP_LIB pointing at the probe build, `ablate=<code>` selects the code
variants compiled side by side there."""
import argparse
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="auto;wgrad_kernel=3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    os.environ.setdefault("RNET_AB_WORKSPACES", "1")
    from bench import synth_ground_truth
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    dev = torch.device("cuda:0")
    params = default_params(input_size=a.size, batch_train=a.batch)
    builder = ModelBuilder(params, "train", device=dev, seed=1337)
    model = builder()
    rx = [builder.FREEZE_VARS_REGEX[n] for n in params.training.freeze_variables]
    eng = TrainEngine(model, a.batch, frozen_regexes=rx, world_size=1)
    enc = LabelEncoder(params, device=dev)
    gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(a.batch, a.size, 1337)]
    images = torch.randn((a.batch, a.size, a.size, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
    base_w = [(p, type(p.opts).from_buffer_copy(p.opts)) for _, p in eng.wgrad_launches]
    base_c = [(p, type(p.opts).from_buffer_copy(p.opts)) for _, p in eng.conv_launches]
    # (kernel selection of the convolutions is fixed at engine construction — BatchNorm partial layouts depend on it —
    # so only fields that keep the kernel family may change there)

    def apply(spec):
        for p, o in base_w + base_c:
            p.opts = o
        if spec == "auto":
            return
        import ctypes
        for kv in spec.split(","):
            k, v = kv.split("=")
            targets = base_c if k.startswith("conv.") else base_w
            only = None
            for pre, kid in (("small.", 0), ("big.", 1), ("halo.", 2)):   # only the weight-gradient problems on that kernel
                if k.startswith(pre):
                    k, only = k[len(pre):], kid
            for p, o in targets:
                if only is not None:
                    q = type(p).from_buffer_copy(p)
                    q.opts = o
                    if eng.lib.rn_wgrad_kernel_id(ctypes.byref(q)) != only:
                        continue
                setattr(p.opts, k.replace("conv.", ""), int(v))

    def step():
        targets = enc.encode_batch(gb, gc, cnt)
        return eng.train_step(images, targets)
    variants = a.variants.split(";")
    for v in variants:      # warm-up, and every variant must train
        apply(v)
        for _ in range(2):
            out = step()
        torch.cuda.synchronize()
        assert np.isfinite(float(out["weighted-loss"].item())), v
    times = {v: [] for v in variants}
    for _ in range(a.rounds):
        for v in variants:
            apply(v)
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / a.steps * 1e3)
    for v in variants:
        t = times[v]
        print(f"{v:40s} median {statistics.median(t):7.3f} ms/step  min {min(t):7.3f}  ({a.batch / statistics.median(t) * 1e3:7.1f} images/s)")


if __name__ == "__main__":
    main()
