"""A/B of the halo patch row pitch (W + 1 against that rounded up to a multiple of 8 pixels = whole LDS bank rows) on
the head-tower conv and on one training step.  python tools/ab_halo_pitch.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import time
import torch
from retinanet import _C
lib = _C.lib()
sys.path.insert(0, os.path.join(ROOT, "tools"))


def step_time(n=12):
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda:0")
    p = default_params(input_size=640, batch_train=32)
    b = ModelBuilder(p, "train", device=dev, seed=1337)
    m = b()
    eng = TrainEngine(m, 32, frozen_regexes=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables], world_size=1)
    enc = LabelEncoder(p, device=dev)
    gb, gc, cnt = [t.to(dev) for t in bench.synth_ground_truth(32, 640, 1337)]
    images = torch.randn((32, 640, 640, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
    for _ in range(3):
        eng.train_step(images, enc.encode_batch(gb, gc, cnt))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.train_step(images, enc.encode_batch(gb, gc, cnt))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    for mode in (0, 1):
        lib.rn_debug_conv_halo_pitch(mode)
        print(f"pitch mode {mode} ({'W+1' if mode == 0 else 'round8(W+1)'}): training step {step_time():.2f} ms", flush=True)
