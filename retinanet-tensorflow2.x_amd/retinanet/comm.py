"""`rn_comm` — the C-ABI collectives (csrc/rn_comm.hip: RCCL on the caller's stream) for the data-parallel step.

torch.distributed stays the bootstrap (rendezvous, the unique-id hand-off) and the fallback; the per-step small
messages — ~65 + 65 SyncBatchNorm all-reduces and the loss normaliser — go through `rn_allreduce_small` when a native
communicator is up: one ncclAllReduce in program order on the compute stream instead of a Python dispatch through
c10d plus a hop to the process group's stream and back per message.

`maybe_enable_native(engine)` is what the executor / bench call when world > 1.  Every fallible stage is COLLECTIVE:
the ranks agree (a MIN over torch.distributed) that librccl loads everywhere before anyone enters the collective
`rn_comm_init`, rank 0 broadcasts (ok, unique id) so a failed draw makes all ranks bail together, the init status is
agreed on before any rank uses or abandons the communicator, and the communicator is VALIDATED against
torch.distributed on a known vector before the engine gets it.  A failure on one rank therefore sends every rank down
the torch.distributed path — never one rank into `except` while the others wait in a collective.

Default since round 3: OFF (RNET_COMM=native turns it on).  Measured on one MI355X with a single-rank communicator
(tools/probes/allreduce_small_cost.py, all a one-GPU box can run): 130 back-to-back 2 KB messages cost 4 - 6 us each on the
stream, but the SyncBatchNorm pattern — a kernel, then a message, 130 times — took 10.6 - 46 ms through rn_allreduce_small
on the compute stream (75 - 350 us per message: the host blocks inside ncclAllReduce while the stream it is handed is
busy) against 1.8 ms through torch.distributed (c10d hands RCCL its own, idle stream and orders it with two events:
7.8 us per message).  Until a multi-GPU node shows otherwise the measured path is the default."""
from __future__ import annotations

import ctypes
import logging
import os

import torch

from retinanet import _C


def _agree(ok, device, group):
    """MIN of a local success flag over the job (torch.distributed): True only if every rank says True."""
    import torch.distributed as dist
    t = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() == 1.0)


class NativeComm:
    """One RCCL communicator behind the C ABI.  Construction is collective and never raises on one rank only:
    `self.ok` is the job-wide verdict (False on every rank if any stage failed anywhere)."""

    def __init__(self, rank, world, device, group=None, handle=None, slot=0, lib=None):
        import torch.distributed as dist
        lib = lib if lib is not None else (handle.lib if handle is not None else _C.lib())
        self._h, self.rank, self.world, self.device, self._lib = None, rank, world, device, lib
        self._owned = handle is None      # a communicator created through an rn_handle lives in its slot ...
        self._handle, self._slot = handle, int(slot)   # ... and close() frees that slot (rn_handle_comm_destroy)
        self.ok, self.error, self._slot_mine = False, "", False
        # stage 0 — local preconditions, agreed on before anything collective in RCCL
        try:
            local = lib.rn_comm_available() == 1
        except Exception as e:   # noqa: BLE001
            local, self.error = False, str(e)
        if not _agree(local, device, group):
            self.error = self.error or "librccl is not loadable on every rank"
            return
        # stage 1 — rank 0 draws the unique id and broadcasts (ok, id): all ranks take the same branch
        n = lib.rn_comm_unique_id_bytes()
        box = [None]
        if rank == 0:
            buf = (ctypes.c_char * n)()
            box = [bytes(buf.raw) if lib.rn_comm_unique_id(buf) == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)       # the out-of-band hand-off of the unique id
        if box[0] is None:
            self.error = "rank 0 could not draw an RCCL unique id"
            return
        uid = (ctypes.c_char * n).from_buffer_copy(box[0])
        # stage 2 — the collective init; its status is agreed on before any rank uses or abandons the communicator
        comm, status = ctypes.c_void_p(), -1
        try:
            with torch.cuda.device(device):
                if handle is not None:
                    status = lib.rn_handle_comm_init(handle.h, int(slot), uid, int(rank), int(world))
                    comm = ctypes.c_void_p(lib.rn_handle_comm(handle.h, int(slot))) if status == 0 else comm
                else:
                    status = lib.rn_comm_init(uid, int(rank), int(world), ctypes.byref(comm))
        except Exception as e:   # noqa: BLE001
            self.error = str(e)
        if status != 0 and not self.error:
            self.error = (lib.rn_last_error() or b"").decode()
        self._h = comm if status == 0 else None
        # only a communicator THIS object put into the handle's slot may be freed by close(): RN_EINVAL because the slot was
        # already taken means it belongs to another NativeComm / engine and may be in use
        self._slot_mine = handle is not None and status == 0
        if not _agree(status == 0, device, group):
            self.error = self.error or "rn_comm_init failed on another rank"
            self.close()
            return
        self.ok = True

    def all_reduce_small(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        _C.check(self._lib.rn_allreduce_small(self._h, t.data_ptr(), t.numel(), _C.current_stream()), "rn_allreduce_small")

    def all_reduce_bucket(self, t):
        dt = {torch.float32: _C.RN_DT_F32, torch.bfloat16: _C.RN_DT_BF16}[t.dtype]
        _C.check(self._lib.rn_allreduce_bucket(self._h, t.data_ptr(), t.numel(), dt, _C.current_stream()),
                 "rn_allreduce_bucket")

    def close(self):
        """Abandon the communicator: destroyed here, not at rn_destroy — after a bootstrap that failed on some rank the
        peers have given up too, and ncclCommDestroy at process exit on such a communicator can hang."""
        if self._h and self._owned:
            self._lib.rn_comm_destroy(self._h)
        elif self._handle is not None and getattr(self, "_slot_mine", False):
            # also when rn_handle_comm_init succeeded here and another rank failed; never when this object's own init
            # failed — the slot is then empty, or holds somebody else's communicator
            self._lib.rn_handle_comm_destroy(self._handle.h, self._slot)
            self._slot_mine = False
        self._h = None
        self.ok = False


def maybe_enable_native(engine):
    """Give `engine` (TrainEngine) a validated native communicator for its small messages; returns it or None.
    Collective over engine.pg: every rank returns a communicator or every rank returns None."""
    import torch.distributed as dist
    mode = os.environ.get("RNET_COMM", "torch")
    if not getattr(engine, "dp_active", engine.world > 1) or mode != "native" or not dist.is_initialized():
        return None
    if dist.get_backend(engine.pg) != "nccl":     # gloo (CPU / one-device functional runs): nothing to take over
        return None
    comm = NativeComm(dist.get_rank(engine.pg), engine.world, engine.dev, engine.pg,
                      handle=getattr(engine, "handle", None), slot=0)
    if not comm.ok:
        logging.warning("rn_comm unavailable, SyncBN messages stay on torch.distributed: %s", comm.error)
        return None
    # stage 3 — validate against torch.distributed before trusting it with SyncBatchNorm statistics
    good, err = False, ""
    try:
        g = torch.Generator(device="cpu").manual_seed(1234 + comm.rank)
        x = torch.randn((4099,), generator=g).to(engine.dev)
        want = x.clone()
        dist.all_reduce(want, group=engine.pg)
        with torch.cuda.device(engine.dev):
            comm.all_reduce_small(x)
            torch.cuda.synchronize()
        good = bool(torch.allclose(x, want, rtol=1e-5, atol=1e-5))
    except Exception as e:   # noqa: BLE001 — the verdict below is still collective
        err = str(e)
    if not _agree(good, engine.dev, engine.pg):
        logging.warning("rn_comm disagrees with torch.distributed on some rank, SyncBN messages stay on "
                        "torch.distributed %s", err)
        comm.close()
        return None
    engine.native_comm = comm
    logging.info("SyncBN / normaliser messages go through rn_comm (RCCL on the compute stream)")
    # C1: a second communicator for the gradient buckets (one communicator per stream that carries collectives): with it
    # a bucket's all-reduce is ONE rn_allreduce_bucket enqueued on the stream that prepared the bucket, in program order —
    # no c10d stream, no event hop (TrainEngine._launch_bucket).  Its construction is collective like the first one's; a
    # failure anywhere leaves the buckets on torch.distributed.
    comm2 = NativeComm(dist.get_rank(engine.pg), engine.world, engine.dev, engine.pg,
                       handle=getattr(engine, "handle", None), slot=1)
    good2 = False
    if comm2.ok:
        try:
            x = torch.arange(1031, dtype=torch.float32, device=engine.dev) * (1 + comm2.rank)
            want = x.clone()
            dist.all_reduce(want, group=engine.pg)
            with torch.cuda.device(engine.dev):
                comm2.all_reduce_bucket(x)
                torch.cuda.synchronize()
            good2 = bool(torch.allclose(x, want, rtol=1e-5, atol=1e-5))
        except Exception:   # noqa: BLE001 — the verdict below is still collective
            good2 = False
    if _agree(comm2.ok and good2, engine.dev, engine.pg):
        engine.native_comm_buckets = comm2
        logging.info("gradient buckets go through rn_allreduce_bucket (RCCL on the bucket's own stream)")
    else:
        if comm2.ok:
            comm2.close()
        engine.native_comm_buckets = None
    return comm
