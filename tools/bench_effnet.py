"""EfficientNet-B3 640x640 (BASELINE config 4) on one GPU: training step and inference timing.
python tools/bench_effnet.py [--batch 32] [--model efficientnet-b3] [--size 640]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--infer-batch", type=int, default=8)
    ap.add_argument("--model", default="efficientnet-b3")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    from bench import synth_ground_truth
    from retinanet.cfg import efficientnet_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    dev = torch.device("cuda:0")
    p = efficientnet_params(a.model, input_size=a.size)
    p.architecture.batch_norm.use_sync = False
    b = ModelBuilder(p, "train", device=dev, seed=1)
    model = b()
    eng = TrainEngine(model, a.batch, frozen_regexes=[])
    enc = LabelEncoder(p, device=dev)
    gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(a.batch, a.size, 1)]
    images = torch.randn((a.batch, a.size, a.size, 3), device=dev)
    print("memory allocated GB", round(torch.cuda.memory_allocated() / 1e9, 2))
    for it in range(a.iters + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.train_step(images, enc.encode_batch(gb, gc, cnt))
        th = time.perf_counter() - t0          # host time to enqueue the step
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if it >= 2:
            print("train step %.1f ms -> %.1f img/s  loss %.4f  (host enqueue %.1f ms)"
                  % (dt * 1e3, a.batch / dt, out["weighted-loss"].item(), th * 1e3))
    # back to back, as bench.py's extra.config4 times it (no device sync between the steps)
    targets = enc.encode_batch(gb, gc, cnt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.iters):
        out = eng.train_step(images, targets)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print("train, %d steps back to back: %.2f ms per step -> %.1f img/s" % (a.iters, dt * 1e3, a.batch / dt))
    del eng
    torch.cuda.empty_cache()
    bi = ModelBuilder(p, "val", device=dev, seed=1)
    mi = bi()
    infer = bi.add_post_processing_stage(mi, capture_graph=True)
    x = torch.randn((a.infer_batch, a.size, a.size, 3), device=dev)
    for _ in range(3):
        infer(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        infer(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("inference batch %d (%s): %.2f ms -> %.1f img/s" % (a.infer_batch, p.inference.mode, dt * 1e3, a.infer_batch / dt))


if __name__ == "__main__":
    main()
