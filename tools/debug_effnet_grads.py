"""Per-tensor gradient comparison (TrainEngine vs RefTrainer) for an EfficientNet RetinaNet — debugging aid.
python tools/debug_effnet_grads.py [model] [size] [batch] [dense|loss]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import test_gpu_efficientnet as T
name = sys.argv[1] if len(sys.argv) > 1 else "efficientnet-b0"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
upstream = sys.argv[4] if len(sys.argv) > 4 else "dense"
fwd, rows, losses = T.run_wiring(torch.device("cuda:0"), name, size, B, upstream)
print("forward relative error: max %.4f" % max(fwd.values()), {k: round(v, 4) for k, v in fwd.items()})
print("losses (engine, restatement):", losses)
zero = T._zero_gradient_betas(name)
for r in sorted(rows):
    print(f"{r[0]:+.4f} {r[1]:8.3f} {r[2]:10.3e} {r[3]}{'  [analytically zero]' if r[3] in zero else ''}")
cs = [r[0] for r in rows if r[3] not in zero]
print("cosine median %.4f min %.4f" % (np.median(cs), min(cs)))
