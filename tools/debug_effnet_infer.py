"""Why does the EfficientNet-B3 serving graph of bench.py's extras return no detections?  (debug helper)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch
from retinanet.cfg import efficientnet_params
from retinanet.model import ModelBuilder

dev = torch.device("cuda:0")
p4 = efficientnet_params("efficientnet-b3", input_size=640)
print("mode", p4.inference.mode, "score_threshold", p4.inference.score_threshold, "policy", p4.floatx.precision if hasattr(p4, "floatx") else None)
bi = ModelBuilder(p4, "val", device=dev, seed=1337)
mi = bi()
x = torch.randn((8, 640, 640, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
gen = torch.Generator().manual_seed(1337)
for k, v in mi.variables.items():
    if k.endswith("/gamma") or k.endswith("/moving_variance"):
        v.copy_((torch.rand(v.shape, generator=gen) * 0.5 + 0.75).to(v.device))
    elif k.endswith("/beta") or k.endswith("/moving_mean"):
        v.copy_((torch.randn(v.shape, generator=gen) * 0.1).to(v.device))
mi._refresh()
preds = mi(x)
lg = torch.cat([preds["class-predictions"][l].reshape(-1) for l in "34567"]).float()
print("logits mean/std/min/max", lg.mean().item(), lg.std().item(), lg.min().item(), lg.max().item(), "finite", torch.isfinite(lg).all().item())
bx = torch.cat([preds["box-predictions"][l].reshape(-1) for l in "34567"]).float()
print("box preds std", bx.std().item(), "finite", torch.isfinite(bx).all().item())
keys = [k for k in mi.variables if "class-head-prediction" in k]
print(keys)
name = "class-head/class-head-prediction-conv2d/"
key = name + ("pointwise_kernel" if name + "pointwise_kernel" in mi.variables else "kernel")
mi.variables[key].mul_(1.0 / max(lg.std().item(), 1e-12))
mi._refresh()
preds = mi(x)
lg = torch.cat([preds["class-predictions"][l].reshape(-1) for l in "34567"]).float()
print("after: logits mean/std/min/max", lg.mean().item(), lg.std().item(), lg.min().item(), lg.max().item())
print("frac > -2.944:", (lg > -2.944).float().mean().item())
for cap in (False, True):
    infer = bi.add_post_processing_stage(mi, capture_graph=cap)
    out = infer(x)
    torch.cuda.synchronize()
    print("capture", cap, "valid", out["valid_detections"].tolist(), "top score", out["scores"].max().item())
