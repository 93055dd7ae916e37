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


def _build(world, rank):
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
    p.training.optimizer.clipnorm = 1e9          # per-replica clipping is the one intentionally non-linear stage
    builder = ModelBuilder(p, "train", device=dev, seed=SEED)
    model = builder()
    per = B // world
    eng = TrainEngine(model, per, frozen_regexes=[], world_size=world)
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
    return {"P": eng.P.cpu().numpy(), "loss": float(out["weighted-loss"].item()),
            "mm": {k: v["mm"].cpu().numpy() for k, v in eng.bn_state.items()}}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, eng, images, targets = _build(world, rank)
    assert eng.sync_bn
    out[rank] = _step(model, eng, images, targets)
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
