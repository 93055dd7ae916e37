"""What ONE rn_allreduce_small costs on the compute stream: a single-rank RCCL communicator
(all a one-GPU box can run), N back-to-back 2 KB messages on one stream — launch + completion of the collective
kernel without any peer traffic, i.e. the floor of the ~130-message SyncBatchNorm chain of a data-parallel step.

    python tools/probes/allreduce_small_cost.py"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))


def main():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from retinanet.comm import NativeComm
    comm = NativeComm(0, 1, dev)
    assert comm.ok, comm.error
    x = torch.randn((512,), device=dev)      # 2 KB: [sum | sum of squares] of a 256-channel layer
    big = torch.randn((64 << 20,), device=dev)
    for n in (1, 130):
        for _ in range(3):
            comm.all_reduce_small(x)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n):
                comm.all_reduce_small(x)
            e1.record()
            torch.cuda.synchronize()
            host = (time.perf_counter() - t0) * 1e6
            best = min(best, e0.elapsed_time(e1) * 1e3)
        print(f"{n:4d} x rn_allreduce_small(2 KB), world 1: {best:8.1f} us on the stream ({best / n:6.2f} us per message), "
              f"host loop {host:8.1f} us")
    # the same chain interleaved with a kernel between the messages (the SyncBN pattern: stats kernel -> all-reduce -> apply)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(130):
        big[:1 << 20].mul_(1.0001)
        comm.all_reduce_small(x)
    e1.record()
    torch.cuda.synchronize()
    t_mix = e0.elapsed_time(e1) * 1e3
    e0.record()
    for _ in range(130):
        big[:1 << 20].mul_(1.0001)
    e1.record()
    torch.cuda.synchronize()
    t_k = e0.elapsed_time(e1) * 1e3
    print(f"130 x (4 MB elementwise kernel + message): {t_mix:8.1f} us; the kernels alone {t_k:8.1f} us -> "
          f"{(t_mix - t_k) / 130:6.2f} us added per message")
    # the same pattern through torch.distributed (c10d's own RCCL communicator and stream hand-off)
    for _ in range(3):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(130):
        big[:1 << 20].mul_(1.0001)
        dist.all_reduce(x)
    e1.record()
    torch.cuda.synchronize()
    print(f"130 x (kernel + torch.distributed.all_reduce): {e0.elapsed_time(e1) * 1e3:8.1f} us")
    # and with the messages on a SIDE stream, ordered by events (what a c10d-style hand-off costs on this runtime)
    side = torch.cuda.Stream(dev)
    ev = [torch.cuda.Event() for _ in range(260)]
    e0.record()
    for i in range(130):
        big[:1 << 20].mul_(1.0001)
        ev[2 * i].record()
        side.wait_event(ev[2 * i])
        with torch.cuda.stream(side):
            comm.all_reduce_small(x)
            ev[2 * i + 1].record()
        torch.cuda.current_stream().wait_event(ev[2 * i + 1])
    e1.record()
    torch.cuda.synchronize()
    print(f"130 x (kernel + message on a side stream, two event waits): {e0.elapsed_time(e1) * 1e3:8.1f} us")
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
