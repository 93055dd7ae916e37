"""Does a high-priority main stream (weight gradients on a normal-priority second stream) shorten the step?
python tools/probes/ab_priority.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from retinanet.cfg import default_params
from retinanet.dataloader import LabelEncoder
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine

dev = torch.device("cuda:0")
B = 32
p = default_params(input_size=640, batch_train=B)
b = ModelBuilder(p, "train", device=dev, seed=1337)
m = b()
eng = TrainEngine(m, B, frozen_regexes=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables], world_size=1)
enc = LabelEncoder(p, device=dev)
gb, gc, cnt = [t.to(dev) for t in bench.synth_ground_truth(B, 640, 1337)]
images = torch.randn((B, 640, 640, 3)).to(dev)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")


def run(stream, steps=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            eng.train_step(images, enc.encode_batch(gb, gc, cnt))
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.train_step(images, enc.encode_batch(gb, gc, cnt))
        stream.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3


hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)
for rep in range(2):
    print("default stream   %.3f ms" % run(torch.cuda.current_stream()))
    print("normal  stream   %.3f ms" % run(lo))
    print("high-pri stream  %.3f ms" % run(hi))
