"""Measured parity of the whole-network INFERENCE forward against oracle/model_ref.py (max / mean error relative to the
tensor's range, fraction of exactly equal outputs), per kernel selection.  python tools/parity_report.py [size] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
from retinanet import _C
from model_ref import RefModel
import test_gpu_model as T

cuda = torch.device("cuda:0")
lib = _C.lib()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def run(tag, tile, randomize=True):
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    p = default_params(input_size=size, balanced=True)
    model = ModelBuilder(p, "val", device=cuda)()
    model.launch_opts = dict(conv_tile=tile)   # rn_launch_opts of this model's engines
    if randomize:
        T._randomize(model, 1)
    images = torch.randn((B, size, size, 3), generator=torch.Generator().manual_seed(1337))
    preds = model(images.to(cuda), training=False)
    torch.cuda.synchronize()
    ref = RefModel(p, model.variables, emulate_bf16=True)(images)
    worst_max = worst_mean = 0.0
    for key in ("box-predictions", "class-predictions"):
        for lv in "34567":
            got, want = preds[key][lv].float().cpu(), ref[key][lv]
            sc = (want - want.mean()).abs().max().item() + 1e-9
            e = (got - want).abs()
            worst_max, worst_mean = max(worst_max, e.max().item() / sc), max(worst_mean, e.mean().item() / sc)
    kids = {}
    eng = model.inference_engine(B)
    import ctypes
    for name, pr in eng.conv_problems.items():
        kids[lib.rn_conv_kernel_id(ctypes.byref(pr))] = kids.get(lib.rn_conv_kernel_id(ctypes.byref(pr)), 0) + 1
    print(f"{tag:40s} size {size} B {B}: max err / range {worst_max:.5f}  mean err / range {worst_mean:.6f}  launches by kernel id {kids}")


run("dispatcher's choice, random BN", 0)
run("forced 256-row kernels, random BN", 2)
run("dispatcher's choice, reference init", 0, randomize=False)
