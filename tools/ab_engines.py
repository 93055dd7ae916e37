"""Same-process A/B of two TrainEngine builds that differ in environment switches read at construction time.
    python tools/ab_engines.py --a "" --b "RNET_SKIP_BN_BIAS_GRAD=1" [--batch 32] [--steps 10] [--rounds 4]
Two engines on one model, 3 warm-up steps each, alternating rounds of `steps` full train steps; prints the per-round
ms / step of both and the medians (the boxes of the pool differ by a few per cent: only same-process pairs rank)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import synth_ground_truth  # noqa: E402


def with_env(spec, fn):
    kv = dict(x.split("=", 1) for x in spec.split(";" if ";" in spec else ",") if x)   # ";" when a value holds commas
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="")
    ap.add_argument("--b", default="")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    dev = torch.device("cuda:0")
    B = a.batch
    params = default_params(input_size=a.size, batch_train=B)
    builder = ModelBuilder(params, "train", device=dev, seed=1337)
    model = builder()
    rx = [builder.FREEZE_VARS_REGEX[n] for n in params.training.freeze_variables]
    enc = LabelEncoder(params, device=dev)
    gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(B, a.size, 1337)]
    images = torch.randn((B, a.size, a.size, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
    specs = {"a": a.a, "b": a.b}
    engs = {k: with_env(v, lambda: TrainEngine(model, B, frozen_regexes=rx, world_size=1)) for k, v in specs.items()}

    def run(k, n):
        def go():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = engs[k].train_step(images, enc.encode_batch(gb, gc, cnt))
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3, out
        return with_env(specs[k], go)
    losses = {}
    for k in engs:
        _, out = run(k, 3)
        losses[k] = float(out["weighted-loss"])
    times = {"a": [], "b": []}
    for _ in range(a.rounds):
        for k in ("a", "b"):
            times[k].append(run(k, a.steps)[0])
    print(json.dumps({"a": specs["a"], "b": specs["b"], "ms_a": [round(v, 3) for v in times["a"]],
                      "ms_b": [round(v, 3) for v in times["b"]], "median_a": round(float(np.median(times["a"])), 3),
                      "median_b": round(float(np.median(times["b"])), 3), "loss_after_warmup": losses}))


if __name__ == "__main__":
    main()
