"""Host time per backward step of the EfficientNet-B3 training engine (which closures are slow to ENQUEUE): wraps every
entry of eng.bwd_steps / eng.fwd_steps with a perf_counter pair for a few steps and prints the largest.
python tools/probes/bwd_host_times.py"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("retinanet-tensorflow2.x_amd", "tests/golden"):
    sys.path.insert(0, os.path.join(ROOT, p))
sys.path.insert(0, ROOT)
import torch
from bench import synth_ground_truth
from retinanet.cfg import efficientnet_params
from retinanet.dataloader import LabelEncoder
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine

dev = torch.device("cuda:0")
p = efficientnet_params("efficientnet-b3", input_size=640)
p.architecture.batch_norm.use_sync = False
model = ModelBuilder(p, "train", device=dev, seed=1)()
eng = TrainEngine(model, 32, frozen_regexes=[])
enc = LabelEncoder(p, device=dev)
gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(32, 640, 1)]
images = torch.randn((32, 640, 640, 3), device=dev)
targets = enc.encode_batch(gb, gc, cnt)
for _ in range(3):
    eng.train_step(images, targets)
torch.cuda.synchronize()
acc = collections.defaultdict(list)


def wrap(lst, tag):
    out = []
    for i, fn in enumerate(lst):
        def w(st, fn=fn, i=i):
            t0 = time.perf_counter()
            r = fn(st)
            acc[(tag, i)].append((time.perf_counter() - t0) * 1e6)
            return r
        for a in ("side", "writes", "name"):
            if hasattr(fn, a):
                setattr(w, a, getattr(fn, a))
        out.append(w)
    return out


orig = list(eng.bwd_steps)
eng.bwd_steps = wrap(eng.bwd_steps, "bwd")
t = time.perf_counter()
for _ in range(4):
    eng.train_step(images, targets)
torch.cuda.synchronize()
print("step ms", (time.perf_counter() - t) / 4 * 1e3, "bwd steps", len(eng.bwd_steps))
rows = sorted(((sum(v[1:]) / len(v[1:]), k) for k, v in acc.items()), reverse=True)
print("host us in backward closures:", round(sum(r[0] for r in rows)))
for us, k in rows[:12]:
    fn = orig[k[1]]
    print(k, round(us, 1), "side" if getattr(fn, "side", False) else "main", fn.__qualname__.split(".")[-1],
          (getattr(fn, "writes", None) or [""])[0][:60], [round(x) for x in acc[k]])
