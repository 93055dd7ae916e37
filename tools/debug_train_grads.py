import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "oracle", "tests", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from test_gpu_train_step import _setup, _cos, _rel
from model_ref import RefTrainer
cuda = torch.device("cuda:0")
size, B, balanced = int(sys.argv[1]), int(sys.argv[3]), sys.argv[2] == "1"
p, model, eng, targets, images = _setup(cuda, size, B, balanced)
ref = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=(len(sys.argv) > 4 and sys.argv[4] == '1'))
out = eng.train_step(images.to(cuda), targets)
torch.cuda.synchronize()
r = ref.step(images, targets["_flat"]["class-targets"].cpu().numpy(), targets["_flat"]["box-targets"].cpu().numpy(),
             float(targets["num-positives"].sum().item()), 0.01)
print({k: (out[k].item(), r["losses"][k]) for k in ("box-loss", "class-loss")})
print("gradnorm", out["gradient-norm"].item(), r["grad_norm"])
for k in eng.train_names:
    got = eng._pview(k, eng.G)
    want = r["clipped_grads"][k]
    if k.endswith("/kernel"):
        c = eng.g.convs[k[:-len("/kernel")]]
        got = got.reshape(c["cout"], c["k"], c["k"], c["cin"]).permute(1, 2, 3, 0)
    got = got.reshape(want.shape).cpu()
    print(f"{k:60s} cos {_cos(got, want):+.4f} rel {_rel(got, want):8.4f} |want| {want.norm().item():.3e} |got| {got.double().norm().item():.3e}")
