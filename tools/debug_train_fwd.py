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
torch.cuda.synchronize()
with torch.no_grad():
    c = ref.backbone(images.double())
    f = ref.fpn(dict(c))
    fb = ref.balance(f) if balanced else f
def nhwc(x): return x.permute(0, 2, 3, 1)
for k, name in (("2", "g1b1_out"), ("3", "g2b1_out"), ("4", "g3b1_out"), ("5", "g4b1_out")):
    print("backbone C" + k, _rel(eng.t[name].float().cpu(), nhwc(c[k])))
for l in "34567":
    print("fpn_out", l, _rel(eng.t["fpn_out" + l].float().cpu(), nhwc(f[l])))
    if balanced:
        print("balanced", l, _rel(eng.bal_out["fpn_out" + l].float().cpu(), nhwc(fb[l])))
# tower by tower for the box head at level 3
with torch.no_grad():
    x = fb["3"]
    for i in range(4):
        x = ref._conv(x, f"box-head/box-head-{i}-conv2d")
        print("box tower raw", i, _rel(eng.raw[f"box-head_t{i}_p3"].float().cpu(), nhwc(x)))
        x = ref._bn(x, f"box-head/box-head-{i}-p3-batch_normalization")
        x = torch.relu(x)
        x = x.to(torch.bfloat16).double()
        print("box tower out", i, _rel(eng.t[f"box-head_t{i}_p3"].float().cpu(), nhwc(x)))
