import os, sys
sys.path.insert(0, "retinanet-tensorflow2.x_amd"); sys.path.insert(0, ".")
import torch
from retinanet.cfg import default_params
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine
dev = torch.device("cuda:0")
p = default_params(input_size=640, batch_train=32)
b = ModelBuilder(p, "train", device=dev, seed=1337)
m = b()
eng = TrainEngine(m, 32, frozen_regexes=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables], world_size=1)
fused = set(eng.bn_bwd_fused)
for pb, _, _, _, _, gops in eng.bn_groups.values():
    for i, o in enumerate(gops):
        if o.get("act") == "relu" and not o.get("residual") and eng._bn_trainable(o):
            print(o["out"], "fused" if o["out"] in fused else "NOT fused", "consumers", eng._consumers(o["out"]), "P", int(pb.seg[i].P), "C", int(pb.seg[i].C))
