"""Every implicit-GEMM launch of one training step (forward + data gradient), one-stream backward, HIP events per launch:
time against a practical bound max(FLOPs / 1.4 PFLOP/s, bytes / 5.5 TB/s) — where is the slack?
RNET_WGRAD_STREAM=0 python tools/conv_gaps.py [--batch 32] [--size 640]"""
import argparse, ctypes, os, sys
os.environ.setdefault("RNET_WGRAD_STREAM", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from retinanet import _C
from retinanet.cfg import default_params
from retinanet.dataloader import LabelEncoder
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
p = default_params(input_size=a.size, batch_train=a.batch)
b = ModelBuilder(p, "train", device=dev, seed=1337)
m = b()
eng = TrainEngine(m, a.batch, frozen_regexes=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables], world_size=1)
enc = LabelEncoder(p, device=dev)
gb, gc, cnt = [t.to(dev) for t in bench.synth_ground_truth(a.batch, a.size, 1337)]
images = torch.randn((a.batch, a.size, a.size, 3)).to(dev)
for _ in range(2):
    eng.train_step(images, enc.encode_batch(gb, gc, cnt))
torch.cuda.synchronize()
names = {id(pp): n for n, pp in eng.conv_launches}
rec = []
lib = eng.lib


def launch(pp, st, what):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _C.check(lib.rn_conv2d_nhwc_fwd(ctypes.byref(pp), st), what)
    e1.record()
    rec.append((id(pp), pp, e0, e1))


eng._launch_conv = launch
for _ in range(a.steps):
    eng.train_step(images, enc.encode_batch(gb, gc, cnt))
torch.cuda.synchronize()
agg = {}
order = []
for pid, pp, e0, e1 in rec:
    if pid not in agg:
        agg[pid] = [pp, 0.0, 0]
        order.append(pid)
    agg[pid][1] += e0.elapsed_time(e1)
    agg[pid][2] += 1
rows = []
for pid in order:
    pp, ms, n = agg[pid]
    us = ms / n * 1e3
    fl, by, _ = eng._conv_meta(pp)
    kid = lib.rn_conv_kernel_id(ctypes.byref(pp))
    s = pp.seg[0]
    bound = max(fl / 1.4e15, by / 5.5e12) * 1e6
    desc = f"{pp.R}x{pp.S}/{pp.stride_h} {s.Cin}->{s.Cout} {s.H}x{s.W} seg{pp.num_segments} k{kid}{' res' if s.residual else ''}{' f32' if pp.out_dtype == _C.RN_DT_F32 else ''}"
    rows.append((us - bound, us, bound, fl, by, names.get(pid, "?"), desc))
tot = sum(r[1] for r in rows)
totb = sum(r[2] for r in rows)
print(f"{len(rows)} launches per step, {tot / 1e3:.2f} ms, practical bound {totb / 1e3:.2f} ms")
for ex, us, bound, fl, by, name, desc in sorted(rows, key=lambda r: -r[0]):
    print(f"{us:8.1f} us  bound {bound:7.1f}  excess {ex:7.1f}  {fl / us / 1e6:7.0f} TF/s {by / us / 1e3:6.0f} GB/s  {name:28s} {desc}")
