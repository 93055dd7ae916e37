"""Why does the plain engine run slower next to a 1-rank nccl group?  python tools/probes/dp_plain_anomaly.py"""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch
import torch.distributed as dist
from bench import synth_ground_truth
from retinanet.cfg import default_params
from retinanet.dataloader import LabelEncoder
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine
from retinanet import _C

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B = 32
params = default_params(input_size=640, batch_train=B)
builder = ModelBuilder(params, "train", device=dev, seed=1337)
model = builder()
rx = [builder.FREEZE_VARS_REGEX[n] for n in params.training.freeze_variables]
enc = LabelEncoder(params, device=dev)
gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(B, 640, 1337)]
images = torch.randn((B, 640, 640, 3), generator=torch.Generator().manual_seed(1337)).to(dev)

def run(eng, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.train_step(images, enc.encode_batch(gb, gc, cnt))
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)

plain = TrainEngine(model, B, frozen_regexes=rx, world_size=1, force_dp=False)
run(plain, 3)
print("plain alone, no process group:", [run(plain) for _ in range(3)], flush=True)
plain8 = TrainEngine(model, B, frozen_regexes=rx, world_size=1, force_dp=False, launch_opts=dict(reserved_cus=8))
run(plain8, 3)
print("plain, reserved_cus=8:", [run(plain8) for _ in range(3)], "plain again:", [run(plain) for _ in range(2)], flush=True)
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
print("plain after init_process_group:", [run(plain) for _ in range(3)], flush=True)
dp = TrainEngine(model, B, frozen_regexes=rx, world_size=1, force_dp=True)
run(dp, 3)
print("dp:", [run(dp) for _ in range(3)], "plain after dp engine exists:", [run(plain) for _ in range(3)], flush=True)
os.environ["RNET_C1_OVERLAP"] = "0"
print("dp, plain order (no overlap):", [run(dp) for _ in range(3)], flush=True)
os.environ.pop("RNET_C1_OVERLAP", None)
os.environ["RNET_COMM"] = "native"
from retinanet import comm
dpn = TrainEngine(model, B, frozen_regexes=rx, world_size=1, force_dp=True)
nat = comm.maybe_enable_native(dpn)
run(dpn, 3)
print("dp, rn_comm (native small messages + native buckets):", nat is not None, dpn.native_comm_buckets is not None,
      [run(dpn) for _ in range(3)], flush=True)
dpn.native_comm = None      # small messages back on torch.distributed, buckets stay native
print("dp, torch small messages + native buckets:", [run(dpn) for _ in range(3)], flush=True)
dist.destroy_process_group()
