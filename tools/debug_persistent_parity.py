"""Whole-network backward parity (cosine vs the bf16-emulating autograd restatement) with the persistent kernels
forced on in different combinations.  python tools/debug_persistent_parity.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch
from retinanet import _C
from model_ref import RefTrainer
import test_gpu_train_step as T

cuda = torch.device("cuda:0")
lib = _C.lib()


def run(tag, conv_tile, wg_big, halo=1, fuse="1"):
    os.environ["RNET_FUSE_BN_STATS"] = fuse
    p, model, eng, targets, images = T._setup(cuda, 256, 4, True, freeze=True,
                                              launch_opts=dict(conv_tile=conv_tile, wgrad_kernel=2 if wg_big else 0,
                                                               conv_no_halo=0 if halo else 1))
    ref = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True)
    preds = eng.forward(images.to(cuda))
    g = torch.Generator().manual_seed(99)
    up = {k: {lv: torch.randn(preds[k][lv].shape, generator=g) for lv in preds[k]} for k in preds}
    eng.backward({k: {lv: t.to(cuda) for lv, t in d.items()} for k, d in up.items()})
    torch.cuda.synchronize()
    rp = ref.forward_train(images)
    fwd = max(T._rel(preds[k][lv].float().cpu(), rp[k][lv].detach()) for k in up for lv in up[k])
    sum((rp[k][lv] * up[k][lv].double()).sum() for k in up for lv in up[k]).backward()
    rows = []
    for k in eng.train_names:
        if k.endswith("/bias") and "prediction" not in k:
            continue
        want = ref.leaf[k].grad
        got = T._engine_grad(eng, k).reshape(want.shape)
        rows.append((T._cos(got, want), k))
    rows.sort()
    cs = [r[0] for r in rows]
    kern = [r[0] for r in rows if r[1].endswith("/kernel")]
    print(f"{tag:38s} fwd rel {fwd:.4f}  cos min {cs[0]:.4f} ({rows[0][1]})  median {np.median(cs):.4f}  kernels median {np.median(kern):.4f} min {min(kern):.4f}")


run("default (128-row kernels)", 0, False)
run("conv persistent, halo, fused stats", 2, False)
run("conv persistent, halo, unfused stats", 2, False, 1, "0")
run("conv persistent, no halo", 2, False, 0)
run("wgrad_big only", 0, True)
run("everything persistent", 2, True)
