"""EfficientNet-B3 640x640 batch-8 serving step (PerClassSoftNMS, HIP-graph replay) with NMS work: random BatchNorm
parameters + class logits rescaled to N(-4.595, 1) like bench.py's `extra.config4.infer`.  python tools/bench_effnet_infer.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch
from retinanet.cfg import efficientnet_params
from retinanet.model import ModelBuilder

dev = torch.device("cuda:0")
p4 = efficientnet_params("efficientnet-b3", input_size=640)
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
std = torch.cat([preds["class-predictions"][l].reshape(-1) for l in "34567"]).float().std().item()
mi.variables["class-head/class-head-prediction-conv2d/pointwise_kernel"].mul_(1.0 / std)
mi._refresh()
infer = bi.add_post_processing_stage(mi, capture_graph=True)
for _ in range(3):
    out = infer(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = infer(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"{dt * 1e3:.2f} ms per batch of 8 -> {8 / dt:.1f} images/s; valid {out['valid_detections'].tolist()}")
