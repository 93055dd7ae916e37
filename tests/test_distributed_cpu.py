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
    # C3: SyncBN: all-reduce [sum, sumsq] of this rank's activations, then finalise with count*world
    x = torch.randn((64, 8), generator=g) * (1 + rank) + rank
    sums = torch.stack([x.sum(0), (x * x).sum(0)])
    dist.all_reduce(sums)
    n = x.shape[0] * world
    mean = sums[0] / n
    var = sums[1] / n - mean * mean
    st = Strategy("multi_gpu", torch.device("cpu"), rank, world)
    gathered = st.gather(x)
    metric = st.reduce_mean(torch.tensor([float(rank)]))
    out[rank] = dict(mine=mine.numpy(), reduced=grads.numpy(), norm=norm.item(), mean=mean.numpy(), var=var.numpy(),
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
