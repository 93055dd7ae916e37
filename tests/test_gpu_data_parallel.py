"""Data-parallel semantics of the whole training step on the GPU: two ranks with half of the
batch each (SyncBN statistics, loss normaliser and gradient SUM through torch.distributed) must land on the
same weights as one process with the whole batch.  A single-GPU box cannot run RCCL with two ranks on one
device, so both ranks use cuda:0 and the collectives go through gloo — the engine code path is the same."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG

pytestmark = pytest.mark.gpu
SIZE, B, SEED = 128, 4, 21


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(world, rank, clipnorm=1e9, force_dp=None):
    sys.path.insert(0, PKG)
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_gt
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    dev = torch.device("cuda:0")
    p = default_params(input_size=SIZE, batch_train=B)
    p.architecture.backbone.depth = 14
    p.training.optimizer.clipnorm = clipnorm     # per-replica clipping is the one intentionally non-linear stage
    builder = ModelBuilder(p, "train", device=dev, seed=SEED)
    model = builder()
    per = B // world
    eng = TrainEngine(model, per, frozen_regexes=[], world_size=world, force_dp=force_dp)
    enc = LabelEncoder(p, device=dev)
    rng = np.random.default_rng(SEED)
    gts = [synth_gt(rng, int(rng.integers(2, 6)), SIZE) for _ in range(B)]
    images = torch.randn((B, SIZE, SIZE, 3), generator=torch.Generator().manual_seed(SEED))
    sl = slice(rank * per, (rank + 1) * per)
    gts = gts[sl]
    Gmax = max(x[0].shape[0] for x in gts)
    gb, gc, cnt = np.zeros([per, Gmax, 4], np.float32), np.zeros([per, Gmax], np.float32), np.zeros([per], np.int32)
    for i, (b, c) in enumerate(gts):
        gb[i, :len(b)], gc[i, :len(c)], cnt[i] = b, c, len(b)
    targets = enc.encode_batch(torch.from_numpy(gb), torch.from_numpy(gc), torch.from_numpy(cnt))
    return model, eng, images[sl].to(dev), targets


def _step(model, eng, images, targets):
    model.optimizer.lr = lambda step: 0.01
    out = eng.train_step(images, targets)
    torch.cuda.synchronize()
    gb = np.zeros(eng.P.numel(), bool)           # which arena elements are BatchNorm gamma / beta
    for k in eng.train_names:
        if k.endswith(("/gamma", "/beta")):
            off, n = eng.p_off[k]
            gb[off:off + n] = True
    return {"P": eng.P.cpu().numpy(), "loss": float(out["weighted-loss"].item()), "gamma_beta": gb,
            "norm_before_clip": float(eng.metrics[1].item()), "clip_fired": bool(getattr(eng, "clip_fired", False)),
            "overlapped": bool(eng._overlap_on), "l2": float(out["l2-regularization"].item()),
            "mm": {k: v["mm"].cpu().numpy() for k, v in eng.bn_state.items()}}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, eng, images, targets = _build(world, rank)
    assert eng.sync_bn
    out[rank] = _step(model, eng, images, targets)
    assert out[rank]["overlapped"]                 # the default N > 1 path: buckets go out during the backward pass
    assert eng.c2_normalizer is not None           # ... and the loss normaliser rode in the first SyncBN message
    dist.destroy_process_group()


def _worker_orders(rank, world, port, out):
    """the same step three ways: plain order (clip, then one all-reduce after the backward pass), overlapped buckets
    without a clip, overlapped buckets with a clip that fires on ONE rank only"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    for tag, overlap, clip in (("plain", "0", 1e9), ("overlapped", "auto", 1e9)):
        os.environ["RNET_C1_OVERLAP"] = overlap
        model, eng, images, targets = _build(world, rank, clipnorm=clip)
        res[tag] = _step(model, eng, images, targets)
        del model, eng
        torch.cuda.empty_cache()
    norms = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(norms, torch.tensor([res["plain"]["norm_before_clip"]]))
    norms = sorted(float(n) for n in norms)
    clip = 0.5 * (norms[0] + norms[1])             # between the two ranks' local gradient norms
    res["clipnorm"] = clip
    for tag, overlap in (("plain_clip", "0"), ("overlapped_clip", "auto")):
        os.environ["RNET_C1_OVERLAP"] = overlap
        model, eng, images, targets = _build(world, rank, clipnorm=clip)
        res[tag] = _step(model, eng, images, targets)
        del model, eng
        torch.cuda.empty_cache()
    os.environ.pop("RNET_C1_OVERLAP", None)
    out[rank] = res
    dist.destroy_process_group()


def test_two_ranks_match_one_process(cuda):
    model, eng, images, targets = _build(1, 0)
    w0 = eng.P.cpu().numpy().copy()
    single = _step(model, eng, images, targets)
    del model, eng
    torch.cuda.empty_cache()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    np.testing.assert_array_equal(r0["P"], r1["P"])                 # replicas stay identical
    upd_1 = single["P"] - w0
    upd_2 = r0["P"] - w0
    # same update up to bf16 noise (per-rank batches change the split-K / reduction order, not the math)
    cos = float(np.dot(upd_1, upd_2) / (np.linalg.norm(upd_1) * np.linalg.norm(upd_2)))
    assert cos > 0.98, cos
    assert abs(np.linalg.norm(upd_2) / np.linalg.norm(upd_1) - 1) < 0.03
    # each rank reports its local sum over (global normaliser / replicas) (retinanet_loss.py:46-49), so the MEAN
    # over replicas (executor.py:450-452) is the single-process loss
    assert 0.5 * (r0["loss"] + r1["loss"]) == pytest.approx(single["loss"], rel=0.02)
    for k in single["mm"]:                                          # SyncBN: global batch statistics
        np.testing.assert_allclose(r0["mm"][k], single["mm"][k], rtol=2e-2, atol=2e-3)
    # BatchNorm gamma / beta on their own (0.1 % of the parameters: invisible in the whole-arena cosine): their
    # gradients are each replica's LOCAL sums until the optimizer's all-reduce (they used to come out
    # world x too large because the SyncBN backward all-reduce ran before they were written)
    gb = single["gamma_beta"]
    assert gb.sum() > 1000
    cos_gb = float(np.dot(upd_1[gb], upd_2[gb]) / (np.linalg.norm(upd_1[gb]) * np.linalg.norm(upd_2[gb])))
    assert cos_gb > 0.98, cos_gb
    assert abs(np.linalg.norm(upd_2[gb]) / np.linalg.norm(upd_1[gb]) - 1) < 0.05
    assert r0["l2"] == pytest.approx(single["l2"], rel=1e-5)        # the l2 term is the same on every replica


def test_overlapped_allreduce_equals_the_plain_order(cuda):
    """gradient buckets all-reduced DURING the backward pass on unclipped gradients, the clip applied
    as a correction only when some rank's factor != 1.  Against the reference's literal order (clip the local
    gradients, then all-reduce — executor.py:432-437): bit-identical weights while no clip fires, and the same
    weights to fp32 rounding when the clip fires on one rank only."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_orders, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        r = out[rank]
        assert not r["plain"]["overlapped"] and r["overlapped"]["overlapped"]
        assert not r["overlapped"]["clip_fired"]
        np.testing.assert_array_equal(r["plain"]["P"], r["overlapped"]["P"])
        assert r["overlapped_clip"]["overlapped"] and r["overlapped_clip"]["clip_fired"]
        np.testing.assert_allclose(r["plain_clip"]["P"], r["overlapped_clip"]["P"], rtol=1e-5, atol=1e-7)
        assert np.abs(r["plain_clip"]["P"] - r["plain"]["P"]).max() > 0      # the clip changed the update
    np.testing.assert_array_equal(out[0]["overlapped_clip"]["P"], out[1]["overlapped_clip"]["P"])
    lo, hi = sorted(out[r]["plain"]["norm_before_clip"] for r in (0, 1))
    assert lo < out[0]["clipnorm"] < hi


def _worker_native_comm(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, PKG)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    from retinanet.comm import NativeComm
    dev = torch.device("cuda:0")
    comm = NativeComm(rank, world, dev)
    assert comm.ok, comm.error
    x = torch.arange(5000, dtype=torch.float32, device=dev) * 0.25
    want = x.clone()
    with torch.cuda.device(dev):
        comm.all_reduce_small(x[:4099])
        y = torch.randn((1 << 20,), device=dev).to(torch.bfloat16)
        y0 = y.clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):             # a collective is enqueued on the CALLER'S stream
            comm.all_reduce_bucket(y)
        torch.cuda.synchronize()
    out[rank] = bool(torch.equal(x, want)) and bool(torch.equal(y, y0))
    comm.close()
    dist.destroy_process_group()


def test_native_comm_single_rank_plumbing(cuda):
    """rn_comm over RCCL with ONE rank (all a one-GPU box can run): dlopen of librccl, the unique-id hand-off through
    torch.distributed, ncclCommInitRank, all-reduces on the current and on a side stream — a sum over one rank is the
    identity.  (Two ranks need two GPUs: RCCL refuses two ranks on one device.)"""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_native_comm, args=(1, _free_port(), out), nprocs=1, join=True)
    assert out[0] is True


def _worker_forced_dp(rank, world, port, out):
    """ONE rank on a real `nccl` (RCCL) process group: the plain step, the data-parallel machinery forced on over
    torch.distributed, and forced on over rn_comm (RNET_COMM=native: SyncBN messages through rn_allreduce_small on the
    compute stream, gradient buckets through rn_allreduce_bucket on the weight-gradient stream)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, PKG)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    from retinanet import comm
    res = {}
    for tag, force, mode in (("plain", False, "torch"), ("dp_torch", True, "torch"), ("dp_native", True, "native")):
        os.environ["RNET_COMM"] = mode
        model, eng, images, targets = _build(1, 0, force_dp=force)
        native = comm.maybe_enable_native(eng)
        r = _step(model, eng, images, targets)
        r["sync_bn"], r["native"] = bool(eng.sync_bn), native is not None
        r["native_buckets"] = getattr(eng, "native_comm_buckets", None) is not None
        r["messages"] = eng.syncbn_messages_per_step
        r["overlap_unsafe"] = bool(getattr(eng, "_overlap_unsafe", False))
        r["side_stream_probed"] = getattr(eng, "side_stream_probed", None)
        res[tag] = r
        if native is not None:
            if eng.native_comm_buckets is not None:
                eng.native_comm_buckets.close()
            native.close()
        del model, eng
        torch.cuda.empty_cache()
    os.environ.pop("RNET_COMM", None)
    out[rank] = res
    dist.destroy_process_group()


def test_forced_data_parallel_path_on_one_rank_of_rccl(cuda):
    """TrainEngine(force_dp=True) on a 1-rank nccl group (what bench.py's extra.dp_overhead times): SyncBN messages, gradient
    buckets prepared and all-reduced while the backward pass runs, the clip-flag read — through torch.distributed and
    through rn_comm (rn_allreduce_small / rn_allreduce_bucket have a product caller: RNET_COMM=native).  A sum over one rank
    is the identity, so all three land on bit-identical weights and the same loss; the side stream came out of the
    queue-independence probe."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_forced_dp, args=(1, _free_port(), out), nprocs=1, join=True)
    r = out[0]
    assert not r["plain"]["sync_bn"] and not r["plain"]["overlapped"] and r["plain"]["messages"] == 0
    for tag in ("dp_torch", "dp_native"):
        assert r[tag]["sync_bn"] and r[tag]["messages"] > 10, (tag, r[tag]["messages"])
        # buckets go out during the backward pass — unless the queue probe found c10d's bucket stream on the main
        # stream's hardware queue in this process: then the job keeps the plain order (and says so)
        assert r[tag]["overlapped"] or r[tag]["overlap_unsafe"], tag
        assert not r[tag]["clip_fired"]
        np.testing.assert_array_equal(r[tag]["P"], r["plain"]["P"])
        assert r[tag]["loss"] == r["plain"]["loss"]
        assert r[tag]["side_stream_probed"] in (True, False)        # probed (None = RNET_STREAM_PROBE=0)
    assert not r["dp_torch"]["native"] and r["dp_native"]["native"] and r["dp_native"]["native_buckets"]
