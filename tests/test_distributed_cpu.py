"""CPU, world_size 2 over gloo: the N>1 host logic of the data-parallel path (C1, C2, C3 of SURVEY
§2.2) — bucketed gradient all-reduce, the loss normaliser, and the SyncBN statistics merge —
checked against the single-process result on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from retinanet.distribute import Strategy, all_reduce_sum_bucketed, global_normalizer
    g = torch.Generator().manual_seed(100 + rank)
    # C1: flat gradient arena in uneven buckets
    grads = torch.randn((100003,), generator=g)
    mine = grads.clone()
    all_reduce_sum_bucketed(grads, world, bucket_bytes=4 * 30000)
    # C2: normaliser
    npos = torch.tensor(float(10 + 5 * rank))
    norm = global_normalizer(npos, world)
    # C3 (+ C2 folded into the step's first message): the engine's own merge helper (TrainEngine._bn_stats_finalize calls
    # the same function) on this rank's [sum | sumsq | spare slot], then the finalisation with count * world
    from retinanet.distribute import syncbn_merge
    x = torch.randn((64, 8), generator=g) * (1 + rank) + rank
    sums = torch.cat([x.sum(0), (x * x).sum(0), torch.zeros(1)])
    norm_folded = syncbn_merge(sums, world, dist.all_reduce, c2_local=(npos + 1.0).reshape(1))
    assert syncbn_merge(sums.clone(), world, dist.all_reduce) is None          # later messages carry no normaliser
    n = x.shape[0] * world
    mean = sums[0:8] / n
    var = sums[8:16] / n - mean * mean
    st = Strategy("multi_gpu", torch.device("cpu"), rank, world)
    gathered = st.gather(x)
    metric = st.reduce_mean(torch.tensor([float(rank)]))
    out[rank] = dict(mine=mine.numpy(), reduced=grads.numpy(), norm=norm.item(), norm_folded=norm_folded.item(),
                     mean=mean.numpy(), var=var.numpy(),
                     x=x.numpy(), gathered=gathered.numpy(), metric=metric.item())
    dist.destroy_process_group()


def test_data_parallel_host_logic_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [out[i] for i in range(world)]
    total = r[0]["mine"] + r[1]["mine"]
    for i in range(world):
        np.testing.assert_allclose(r[i]["reduced"], total, rtol=1e-6, atol=1e-6)
        assert r[i]["norm"] == ((10 + 1) + (15 + 1)) / 2      # retinanet_loss.py:38-49
        assert r[i]["norm_folded"] == r[i]["norm"]            # the same scalar when it rides in the first SyncBN message
        assert r[i]["metric"] == 0.5
    allx = np.concatenate([r[0]["x"], r[1]["x"]])
    np.testing.assert_allclose(r[0]["mean"], allx.mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r[0]["var"], allx.var(0), rtol=1e-4, atol=1e-5)   # biased variance (SyncBN)
    np.testing.assert_array_equal(r[1]["gathered"], allx)


def test_get_strategy_types():
    sys.path.insert(0, PKG)
    import pytest
    from retinanet.cfg import AttrDict
    from retinanet.distribute import get_strategy
    with pytest.raises(ValueError):
        get_strategy(AttrDict(type="tpu", name=""))
    with pytest.raises(ValueError):
        get_strategy(AttrDict(type="bogus", name=""))
    s = get_strategy(AttrDict(type="gpu", name=""))
    assert s.num_replicas_in_sync == 1


class _FakeCommLib:
    """Stands in for librnet_hip.so's rn_comm entry points so that a failure can be injected on ONE rank at each stage
    of NativeComm's construction (no GPU, no RCCL: the point is the control flow around the collectives)."""

    def __init__(self, rank, fail_stage, fail_rank):
        self.rank, self.fail_stage, self.fail_rank = rank, fail_stage, fail_rank
        self.inits = 0
        self.slots, self.destroyed = {}, 0

    def _fails(self, stage):
        return self.fail_stage == stage and self.rank == self.fail_rank

    def rn_comm_available(self):
        return 0 if self._fails("available") else 1

    def rn_comm_unique_id_bytes(self):
        return 128

    def rn_comm_unique_id(self, buf):
        return -4 if self._fails("unique_id") else 0

    def rn_comm_init(self, uid, rank, world, comm_out):
        self.inits += 1
        if self._fails("init"):
            return -4
        comm_out._obj.value = 0x1000 + rank
        return 0

    def rn_comm_destroy(self, comm):
        return 0

    def rn_last_error(self):
        return b"injected failure"

    # the same communicator created through an rn_handle: it lives in a slot of the handle
    def rn_handle_comm_init(self, h, slot, uid, rank, world):
        self.inits += 1
        if self._fails("init"):
            return -4
        assert self.slots.get(slot) is None, "slot already holds a communicator"
        self.slots[slot] = 0x2000 + rank
        return 0

    def rn_handle_comm(self, h, slot):
        return self.slots.get(slot)

    def rn_handle_comm_destroy(self, h, slot):
        self.destroyed += int(self.slots.pop(slot, None) is not None)
        return 0


def _worker_comm_failure(rank, world, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import contextlib
    from unittest import mock
    from retinanet.comm import NativeComm
    res = {}
    # (stage, rank that fails): a failed draw can only happen on rank 0; the others on either rank
    cases = [("none", 0), ("available", 0), ("available", 1), ("unique_id", 0), ("init", 0), ("init", 1)]
    with mock.patch("torch.cuda.device", lambda d: contextlib.nullcontext()):   # CPU box: no device to select
        for stage, who in cases:
            lib = _FakeCommLib(rank, stage, who)
            comm = NativeComm(rank, world, torch.device("cpu"), None, lib=lib)
            res[f"{stage}@{who}"] = (comm.ok, lib.inits, comm._h is not None)
        # through an rn_handle (what Executor / bench do): after a failure on EITHER rank the slot is empty on BOTH — the
        # survivor's communicator is destroyed now, a later rn_handle_comm_init finds the slot free
        from types import SimpleNamespace
        for stage, who in [("none", 0), ("init", 0), ("init", 1)]:
            lib = _FakeCommLib(rank, stage, who)
            comm = NativeComm(rank, world, torch.device("cpu"), None, handle=SimpleNamespace(h=1, lib=lib), slot=2)
            res[f"handle:{stage}@{who}"] = (comm.ok, lib.slots.get(2) is not None, lib.destroyed)
            if comm.ok:
                comm.close()
                res[f"handle:{stage}@{who}:closed"] = (lib.slots.get(2) is not None, lib.destroyed)
    # the ranks are still in lockstep: one more collective completes
    t = torch.ones(1)
    dist.all_reduce(t)
    res["tail"] = t.item()
    out[rank] = res
    dist.destroy_process_group()


def test_native_comm_construction_is_collective_safe():
    """retinanet/comm.py: a failure on ONE rank at any stage — librccl not loadable,
    the unique-id draw, the collective init itself — must send EVERY rank down the fallback together, never one rank into
    an exception while the others wait in a collective.  Two gloo ranks, injected failures; the run deadlocks (and the
    test times out) if a stage is not collective."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_comm_failure, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["tail"] == r1["tail"] == 2.0
    assert r0["none@0"] == r1["none@0"] == (True, 1, True)
    for key in ("available@0", "available@1", "unique_id@0"):
        # agreed on before anyone entered the collective init: no rank called it
        assert r0[key] == r1[key] == (False, 0, False), (key, r0[key], r1[key])
    for key in ("init@0", "init@1"):
        # every rank entered the init; the status is agreed on afterwards and the survivor drops its communicator
        assert r0[key] == r1[key] == (False, 1, False), (key, r0[key], r1[key])
    assert r0["handle:none@0"] == r1["handle:none@0"] == (True, True, 0)
    assert r0["handle:none@0:closed"] == r1["handle:none@0:closed"] == (False, 1)
    for who in (0, 1):
        key = f"handle:init@{who}"
        survivor, failed = (r1, r0) if who == 0 else (r0, r1)
        assert failed[key] == (False, False, 0), (key, failed[key])        # its init failed: the slot was never filled
        assert survivor[key] == (False, False, 1), (key, survivor[key])    # created, then destroyed when the job bailed


def _worker_eval_loop(rank, world, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from retinanet.cfg import AttrDict
    from retinanet.distribute import Strategy
    from retinanet.executor import Executor
    H = W = 8
    per_replica = 3
    # rank 0: 2 full batches + a short one (7 records); rank 1: ONE short batch (2 records) — the shards of a sharded
    # validation set never line up
    counts = {0: [3, 3, 1], 1: [2]}[rank]
    first_id = {0: 100, 1: 200}[rank]

    def dataset():
        nxt = first_id
        for n in counts:
            ids = torch.arange(nxt, nxt + n)
            nxt += n
            img = ids.to(torch.float32).reshape(n, 1, 1, 1).expand(n, H, W, 3).clone()
            yield {"image": img, "image_id": ids, "resize_scale": torch.full((n, 2), 0.5)}

    def fake_model(images, training=False):        # detections that carry the image's id back
        b = images.shape[0]
        v = images[:, 0, 0, 0]
        return {"boxes": v.reshape(b, 1, 1).expand(b, 1, 4).clone(), "scores": v.reshape(b, 1).clone(),
                "classes": torch.zeros((b, 1), dtype=torch.int32), "valid_detections": torch.ones((b,), dtype=torch.int32)}

    ex = Executor.__new__(Executor)
    ex.params = AttrDict(input=AttrDict(input_shape=[H, W]))
    ex.distribute_strategy = Strategy("multi_gpu", torch.device("cpu"), rank, world)
    ex.num_replicas = world
    ex.batch_size = {"val": per_replica * world}
    ex._model = SimpleNamespace(device=torch.device("cpu"))
    ex._eval_model = fake_model
    ex._val_dataset = dataset
    got_ids, got_scores, steps = [], [], 0
    for res in ex._gathered_eval_results(total_steps=10):
        steps += 1
        got_ids += res["image_id"].tolist()
        got_scores += res["detections"]["scores"].reshape(-1).tolist()
        assert res["resize_scale"].shape == (len(res["image_id"]), 2)
    scores = ex.distribute_strategy.broadcast_object({"AP": 0.5} if rank == 0 else None, src=0)
    out[rank] = dict(ids=got_ids, scores=got_scores, steps=steps, ap=scores["AP"])
    dist.destroy_process_group()


def test_multi_replica_evaluation_with_uneven_shards_does_not_hang_or_drop_records():
    """with one input pipeline per rank the validation shards are uneven — a rank that runs out of
    records (or holds a short last batch) must keep taking part in the gathers.  Loop control is collective
    (Strategy.any_true), every rank contributes a fixed-size padded batch + a row mask, the padding rows are dropped
    after the gather: both ranks see every record exactly once, in the same order, and leave the loop together."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_eval_loop, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["ids"] == r1["ids"] and r0["steps"] == r1["steps"] == 3
    assert sorted(r0["ids"]) == [100, 101, 102, 103, 104, 105, 106, 200, 201]
    assert r0["scores"] == [float(i) for i in r0["ids"]]       # the detections travelled with their rows
    assert r0["ap"] == r1["ap"] == 0.5
