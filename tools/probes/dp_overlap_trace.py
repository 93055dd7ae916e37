"""dp engine (forced, 1-rank nccl) for a kernel trace: python tools/probes/dp_overlap_trace.py [overlap|plainorder|plain]"""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
mode = sys.argv[1] if len(sys.argv) > 1 else "overlap"
if mode == "plainorder":
    os.environ["RNET_C1_OVERLAP"] = "0"
import torch
import torch.distributed as dist
from bench import synth_ground_truth
from retinanet.cfg import default_params
from retinanet.dataloader import LabelEncoder
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B = 32
params = default_params(input_size=640, batch_train=B)
builder = ModelBuilder(params, "train", device=dev, seed=1337)
model = builder()
rx = [builder.FREEZE_VARS_REGEX[n] for n in params.training.freeze_variables]
enc = LabelEncoder(params, device=dev)
gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(B, 640, 1337)]
images = torch.randn((B, 640, 640, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
eng = TrainEngine(model, B, frozen_regexes=rx, world_size=1, force_dp=(mode != "plain"))
for _ in range(3):
    eng.train_step(images, enc.encode_batch(gb, gc, cnt))
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 8
for _ in range(n):
    eng.train_step(images, enc.encode_batch(gb, gc, cnt))
torch.cuda.synchronize()
print(mode, "ms/step", round((time.perf_counter() - t0) / n * 1e3, 3), flush=True)
dist.destroy_process_group()
