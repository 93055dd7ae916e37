"""Quick timing of the training step (per-phase, HIP events).  python tools/bench_train.py --batch 32"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch


def synth_targets(enc, B, size, seed):
    rng = np.random.default_rng(seed)
    Gmax = 32
    gb, gc, cnt = np.zeros([B, Gmax, 4], np.float32), np.zeros([B, Gmax], np.float32), np.zeros([B], np.int32)
    for i in range(B):
        G = int(rng.integers(1, 33))
        c = rng.uniform(0, size, (G, 2))
        wh = np.exp(rng.uniform(np.log(8), np.log(512), (G, 2)))
        x1, x2 = np.clip(c - wh / 2, 0, size), np.clip(c + wh / 2, 0, size)
        gb[i, :G] = np.concatenate([(x1 + x2) / 2, np.maximum(x2 - x1, 1.0)], 1)
        gc[i, :G] = rng.integers(0, 80, G)
        cnt[i] = G
    return torch.from_numpy(gb), torch.from_numpy(gc), torch.from_numpy(cnt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--big-min-tiles", type=int, default=0, help="override the 256x256 kernel's minimum tile count")
    ap.add_argument("--wgrad-big-blocks", type=int, default=0)
    ap.add_argument("--wgrad-blocks", type=int, default=0, help="override the wgrad split-K target workgroup count")
    a = ap.parse_args()
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    dev = torch.device("cuda:0")
    opts = dict(conv_big_min_tiles=a.big_min_tiles or 0, wgrad_target_blocks=(a.wgrad_big_blocks or a.wgrad_blocks or 0))
    p = default_params(input_size=a.size)
    b = ModelBuilder(p, "train", device=dev)
    model = b()
    eng = TrainEngine(model, a.batch, frozen_regexes=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables],
                      launch_opts=opts)
    enc = LabelEncoder(p, device=dev)
    gb, gc, cnt = [t.to(dev) for t in synth_targets(enc, a.batch, a.size, 1337)]
    images = torch.randn((a.batch, a.size, a.size, 3), device=dev)
    print("memory allocated GB", torch.cuda.memory_allocated() / 1e9)

    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    for it in range(a.iters + 2):
        e0 = ev()
        targets = enc.encode_batch(gb, gc, cnt)
        e1 = ev()
        preds = eng.forward(images)
        e2 = ev()
        loss = model.loss(targets, preds, compute_grads=True, grad_scale=1.0)
        e3 = ev()
        eng.backward(model.loss.grads)
        e4 = ev()
        eng.optimizer_step(0.01, 0.9, 10.0, 1e-4, 0.9998)
        e5 = ev()
        torch.cuda.synchronize()
        if it >= 2:
            t = [x.elapsed_time(y) for x, y in ((e0, e1), (e1, e2), (e2, e3), (e3, e4), (e4, e5))]
            print(f"encode {t[0]:.2f} fwd {t[1]:.2f} loss {t[2]:.2f} bwd {t[3]:.2f} opt {t[4]:.2f} total {sum(t):.2f} ms "
                  f"-> {a.batch / sum(t) * 1e3:.1f} img/s  loss {loss['weighted-loss'].item():.4f}")


if __name__ == "__main__":
    main()
