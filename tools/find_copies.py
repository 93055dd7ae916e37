"""Which host calls launch the small device copies of a training step?  torch.profiler over two steps, CPU-side op
counts with their Python callers.  python tools/find_copies.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from retinanet.cfg import default_params
from retinanet.dataloader import LabelEncoder
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine

dev = torch.device("cuda:0")
B = 8
p = default_params(input_size=640, batch_train=B)
b = ModelBuilder(p, "train", device=dev, seed=1337)
m = b()
eng = TrainEngine(m, B, frozen_regexes=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables], world_size=1)
enc = LabelEncoder(p, device=dev)
gb, gc, cnt = [t.to(dev) for t in bench.synth_ground_truth(B, 640, 1337)]
images = torch.randn((B, 640, 640, 3)).to(dev)
for _ in range(2):
    eng.train_step(images, enc.encode_batch(gb, gc, cnt))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        eng.train_step(images, enc.encode_batch(gb, gc, cnt))
    torch.cuda.synchronize()
ops = collections.Counter()
stacks = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") or "Memcpy" in e.name or "memcpy" in e.name:
        ops[e.name] += 1
        if e.name in ("aten::copy_", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul", "aten::add", "aten::div"):
            st = [s for s in (e.stack or []) if "retinanet" in s or "bench" in s]
            stacks[(e.name, st[0] if st else "?")] += 1
for k, v in ops.most_common(25):
    print(f"{v / 2:8.1f} per step  {k}")
print()
for k, v in stacks.most_common(30):
    print(f"{v / 2:8.1f} per step  {k[0]:16s} {k[1]}")
