"""Per-launch-group breakdown of one training step: every forward / backward step closure of the TrainEngine is
bracketed with HIP events; conv / wgrad problems are recognised from the closure's bound arguments and priced
(TFLOP/s, algorithmic GB/s).   python tools/step_breakdown.py --batch 32 [--top 40]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from bench_train import synth_targets


def describe(f):
    from retinanet import _C
    objs = list(f.__defaults__ or ())
    for o in list(objs):
        if isinstance(o, tuple):
            objs += list(o)
    for o in objs:
        tgt = getattr(o, "_obj", o)           # ctypes.byref(...) keeps the structure in _obj
        if isinstance(tgt, _C.ConvProblem):
            p = tgt
            fl = by = 0
            for i in range(p.num_segments):
                s = p.seg[i]
                fl += 2 * s.N * s.Ho * s.Wo * p.R * p.S * s.Cin * s.Cout
                by += 2 * s.N * s.H * s.W * s.pix_stride + (4 if p.out_dtype == 0 else 2) * s.N * s.Ho * s.Wo * s.Cout
            s = p.seg[0]
            tile = _C.lib().rn_conv_tile_rows(ctypes.byref(p))
            return (f"conv{p.R}x{p.S}/{p.stride_h} {s.Cin}->{s.Cout} {s.H}x{s.W} seg{p.num_segments} tile{tile}", fl, by)
        if isinstance(tgt, _C.WgradProblem):
            p = tgt
            fl = by = 0
            for i in range(p.num_segments):
                s = p.seg[i]
                fl += 2 * s.N * s.Ho * s.Wo * p.R * p.S * s.Cin * s.Cout
                by += 2 * s.N * s.H * s.W * s.Cin + 2 * s.N * s.Ho * s.Wo * s.Cout
            s = p.seg[0]
            return (f"wgrad{p.R}x{p.S}/{p.stride_h} {s.Cin}->{s.Cout} {s.H}x{s.W} seg{p.num_segments}", fl, by)
    code = f.__code__
    return (f"{code.co_name}:{code.co_firstlineno}", 0, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--wgrad-big-blocks", type=int, default=0)
    ap.add_argument("--only", default="", help="substring filter on the printed names")
    a = ap.parse_args()
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    from retinanet.optimizers import build_optimizer
    dev = torch.device("cuda:0")
    p = default_params(input_size=a.size)
    b = ModelBuilder(p, "train", device=dev)
    model = b()
    model.optimizer = build_optimizer(p.training.optimizer, p.training.train_steps, p.floatx.precision)
    eng = TrainEngine(model, a.batch, frozen_regexes=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables],
                      launch_opts=dict(wgrad_target_blocks=a.wgrad_big_blocks or 0))
    enc = LabelEncoder(p, device=dev)
    gb, gc, cnt = [t.to(dev) for t in synth_targets(enc, a.batch, a.size, 1337)]
    targets = enc.encode_batch(gb, gc, cnt)
    images = torch.randn((a.batch, a.size, a.size, 3), device=dev)
    for _ in range(2):
        eng.train_step(images, targets)
    rec = []

    def wrap(f, phase):
        label = describe(f)

        def g(st):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(st); e1.record()
            rec.append((phase, label, e0, e1))
        return g
    eng.fwd_steps = [wrap(f, "fwd") for f in eng.fwd_steps]
    eng.bwd_steps = [wrap(f, "bwd") for f in eng.bwd_steps]
    eng.train_step(images, targets)
    torch.cuda.synchronize()
    rows = [(ph, lb[0], e0.elapsed_time(e1), lb[1], lb[2]) for ph, lb, e0, e1 in rec]
    tot = sum(r[2] for r in rows)
    print(f"steps {len(rows)} total {tot:.2f} ms (event-bracketed: includes launch gaps of each group)")
    agg = {}
    for ph, name, ms, fl, by in rows:
        k = (ph, name)
        v = agg.setdefault(k, [0.0, 0, 0, 0])
        v[0] += ms; v[1] += fl; v[2] += by; v[3] += 1
    items = [kv for kv in agg.items() if a.only in kv[0][1]]
    print(f"filtered total {sum(v[0] for _, v in items):.3f} ms")
    for (ph, name), (ms, fl, by, n) in sorted(items, key=lambda kv: -kv[1][0])[:a.top]:
        extra = f"{fl / ms / 1e9:7.0f} TF/s {by / ms / 1e6:6.0f} GB/s" if fl else ""
        print(f"{ms:7.3f} ms x{n:2d} {ph} {name:60s} {extra}")


if __name__ == "__main__":
    main()
