"""`rn_comm` — the C-ABI collectives (csrc/rn_comm.hip: RCCL on the caller's stream) for the data-parallel step.

torch.distributed stays the bootstrap (rendezvous, the unique-id hand-off) and the fallback; the per-step small
messages — ~65 + 65 SyncBatchNorm all-reduces and the loss normaliser — go through `rn_allreduce_small` when a native
communicator is up: one ncclAllReduce in program order on the compute stream instead of a Python dispatch through
c10d plus a hop to the process group's stream and back per message.

`maybe_enable_native(engine)` is what the executor / bench call when world > 1: it builds a communicator, VALIDATES it
against torch.distributed on a known vector, and only then hands it to the engine.  RNET_COMM=torch switches it off."""
from __future__ import annotations

import ctypes
import logging
import os

import torch

from retinanet import _C


class NativeComm:
    def __init__(self, rank, world, device, group=None):
        import torch.distributed as dist
        lib = _C.lib()
        n = lib.rn_comm_unique_id_bytes()
        buf = (ctypes.c_char * n)()
        if rank == 0:
            _C.check(lib.rn_comm_unique_id(buf), "rn_comm_unique_id")
        box = [bytes(buf.raw) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)       # the out-of-band hand-off of the unique id
        uid = (ctypes.c_char * n).from_buffer_copy(box[0])
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _C.check(lib.rn_comm_init(uid, int(rank), int(world), ctypes.byref(handle)), "rn_comm_init")
        self._h, self.rank, self.world, self.device, self._lib = handle, rank, world, device, lib

    def all_reduce_small(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        _C.check(self._lib.rn_allreduce_small(self._h, t.data_ptr(), t.numel(), _C.current_stream()), "rn_allreduce_small")

    def all_reduce_bucket(self, t):
        dt = {torch.float32: _C.RN_DT_F32, torch.bfloat16: _C.RN_DT_BF16}[t.dtype]
        _C.check(self._lib.rn_allreduce_bucket(self._h, t.data_ptr(), t.numel(), dt, _C.current_stream()),
                 "rn_allreduce_bucket")

    def close(self):
        if self._h:
            self._lib.rn_comm_destroy(self._h)
            self._h = None


def maybe_enable_native(engine):
    """Give `engine` (TrainEngine) a validated native communicator for its small messages; returns it or None."""
    import torch.distributed as dist
    mode = os.environ.get("RNET_COMM", "auto")
    if engine.world <= 1 or mode == "torch" or not dist.is_initialized():
        return None
    if dist.get_backend(engine.pg) != "nccl":     # gloo (CPU / one-device functional runs): nothing to take over
        return None
    try:
        comm = NativeComm(dist.get_rank(engine.pg), engine.world, engine.dev, engine.pg)
        # validate against torch.distributed before trusting it with SyncBatchNorm statistics
        g = torch.Generator(device="cpu").manual_seed(1234 + comm.rank)
        x = torch.randn((4099,), generator=g).to(engine.dev)
        want = x.clone()
        dist.all_reduce(want, group=engine.pg)
        with torch.cuda.device(engine.dev):
            comm.all_reduce_small(x)
            torch.cuda.synchronize()
        ok = torch.tensor([1.0 if torch.allclose(x, want, rtol=1e-5, atol=1e-5) else 0.0], device=engine.dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=engine.pg)
        if ok.item() != 1.0:
            raise RuntimeError("native all-reduce disagrees with torch.distributed")
    except Exception as e:      # any failure: stay on torch.distributed (all ranks take the same branch: the
        logging.warning("rn_comm unavailable, SyncBN messages stay on torch.distributed: %s", e)   # check is collective)
        return None
    engine.native_comm = comm
    logging.info("SyncBN / normaliser messages go through rn_comm (RCCL on the compute stream)")
    return comm
