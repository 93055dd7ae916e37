"""Which hardware queue does c10d's internal NCCL stream share?  For 12 pool streams X: spin 1 ms on X, then time a sync
all_reduce issued from the main stream.  python tools/probes/c10d_stream_probe.py"""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch
import torch.distributed as dist
from retinanet import _C

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _C.lib()
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
t = torch.ones((512,), device=dev)
dist.all_reduce(t)
g2 = dist.new_group(backend="nccl")
dist.all_reduce(t, group=g2)
torch.cuda.synchronize()
main = torch.cuda.current_stream(dev)
streams = [torch.cuda.Stream(dev) for _ in range(12)]
print("torch", torch.__version__)
for i, x in enumerate(streams):
    row = []
    for grp, name in ((None, "world"), (g2, "g2")):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _C.check(lib.rn_probe_spin(1000, _C.c_void_p(x.cuda_stream)))
        e0.record(main)
        dist.all_reduce(t, group=grp)
        e1.record(main)
        torch.cuda.synchronize()
        row.append(f"{name} sync {e0.elapsed_time(e1) * 1e3:7.0f} us")
        # async from stream x's context, then a tiny kernel on main
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _C.check(lib.rn_probe_spin(1000, _C.c_void_p(x.cuda_stream)))
        with torch.cuda.stream(x):
            w = dist.all_reduce(t, group=grp, async_op=True)
        e0.record(main)
        _C.check(lib.rn_probe_spin(1, _C.c_void_p(main.cuda_stream)))
        e1.record(main)
        with torch.cuda.stream(x):
            w.wait()
        torch.cuda.synchronize()
        row.append(f"async-from-x, main kernel {e0.elapsed_time(e1) * 1e3:7.0f} us")
    ov = "-"
    print(f"stream {i:2d} overlaps main {ov}: " + " | ".join(row), flush=True)
dist.destroy_process_group()
