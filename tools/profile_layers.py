"""Per-launch timing of the inference engine (HIP events, eager): name, ms, GFLOP, TFLOP/s.
Usage (GPU box): python tools/profile_layers.py [--batch 8] [--size 640] [--iters 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import conv_flops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from retinanet import _C
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    dev = torch.device("cuda:0")
    p = default_params(input_size=a.size)
    model = ModelBuilder(p, "val", device=dev)()
    eng = model.inference_engine(a.batch)
    eng.t["images"].normal_()
    st = _C.current_stream()
    for fn, _ in eng.steps:
        fn(st)
    torch.cuda.synchronize()
    tot = {}
    for _ in range(a.iters):
        evs = []
        for fn, name in eng.steps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(st); e1.record()
            evs.append((name, e0, e1))
        torch.cuda.synchronize()
        for i, (name, e0, e1) in enumerate(evs):
            tot[(i, name)] = tot.get((i, name), 0.0) + e0.elapsed_time(e1)
    total_ms = 0.0
    total_fl = 0.0
    print(f"{'step':40s} {'ms':>9s} {'GFLOP':>9s} {'TFLOP/s':>9s}")
    for (i, name), ms in sorted(tot.items()):
        ms /= a.iters
        fl = conv_flops(eng, name) if name.startswith("conv:") and name != "conv:stem" else 0
        if name.startswith("bneck:"):
            fl = next(fb.flops for fb in eng.bneck.values() if fb.name == name)
        if name == "conv:stem":
            fl = 2 * a.batch * (a.size // 2) ** 2 * 147 * 64
        total_ms += ms
        total_fl += fl
        print(f"{name:40s} {ms:9.4f} {fl / 1e9:9.2f} {fl / 1e9 / ms if ms > 0 else 0:9.1f}")
    print(f"{'TOTAL':40s} {total_ms:9.3f} {total_fl / 1e9:9.1f} {total_fl / 1e9 / total_ms:9.1f}")


if __name__ == "__main__":
    main()
