import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "oracle", "tests", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from test_gpu_train_step import _setup, _cos, _rel
from model_ref import RefTrainer
cuda = torch.device("cuda:0")
size, balanced, B = int(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3])
p, model, eng, targets, images = _setup(cuda, size, B, balanced)
ref = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True)
preds = eng.forward(images.to(cuda))
g = torch.Generator().manual_seed(99)
up = {k: {lv: torch.randn(preds[k][lv].shape, generator=g) for lv in preds[k]} for k in preds}
eng.backward({k: {lv: t.to(cuda) for lv, t in d.items()} for k, d in up.items()})
torch.cuda.synchronize()
rp = ref.forward_train(images)
for k in ("class-predictions", "box-predictions"):
    for lv in "34567":
        print(k, lv, "fwd rel err", _rel(preds[k][lv].float().cpu(), rp[k][lv].detach()))
L = sum((rp[k][lv] * up[k][lv].double()).sum() for k in up for lv in up[k])
L.backward()
for k in eng.train_names:
    got = eng._pview(k, eng.G)
    want = ref.leaf[k].grad
    if k.endswith("/kernel"):
        c = eng.g.convs[k[:-len("/kernel")]]
        got = got.reshape(c["cout"], c["k"], c["k"], c["cin"]).permute(1, 2, 3, 0)
    got = got.reshape(want.shape).cpu()
    print(f"{k:60s} cos {_cos(got, want):+.4f} rel {_rel(got, want):8.4f} |want| {want.norm().item():.3e} |got| {got.double().norm().item():.3e}")
