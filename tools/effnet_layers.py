"""Per-launch table of the EfficientNet-B3 training step's implicit-GEMM launches (forward, data gradient, weight gradient)
from TrainEngine.layer_profile (HIP events on the launch stream, one-stream backward so that durations are the kernels' own).
python tools/effnet_layers.py [--batch 32]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
os.environ.setdefault("RNET_WGRAD_STREAM", "0")
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    from bench import synth_ground_truth
    from retinanet.cfg import efficientnet_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    dev = torch.device("cuda:0")
    p = efficientnet_params("efficientnet-b3", input_size=640)
    p.architecture.batch_norm.use_sync = False
    model = ModelBuilder(p, "train", device=dev, seed=1)()
    eng = TrainEngine(model, a.batch, frozen_regexes=[])
    enc = LabelEncoder(p, device=dev)
    gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(a.batch, 640, 1)]
    images = torch.randn((a.batch, 640, 640, 3), device=dev)
    for _ in range(3):
        eng.train_step(images, enc.encode_batch(gb, gc, cnt))
    torch.cuda.synchronize()
    prof = []
    eng.layer_profile = prof
    for _ in range(3):
        eng.train_step(images, enc.encode_batch(gb, gc, cnt))
    eng.layer_profile = None
    torch.cuda.synchronize()
    rows = {}
    for e0, e1, name, fl, by, kern in prof:
        r = rows.setdefault(name, [0.0, 0, fl, by, kern])
        r[0] += e0.elapsed_time(e1) * 1e3; r[1] += 1
    tab = sorted(((v[0] / v[1], k, v) for k, v in rows.items()), reverse=True)
    tot = sum(t for t, _, _ in tab)
    print(f"{len(tab)} launches, {tot / 1e3:.2f} ms per step in implicit-GEMM launches")
    for us, name, v in tab[:a.top]:
        print(f"{name[:58]:58s} {v[4][:40]:40s} {us:8.1f} us  {v[2] / us / 1e6:7.1f} TFLOP/s {v[3] / us / 1e3:7.0f} GB/s")


if __name__ == "__main__":
    main()
