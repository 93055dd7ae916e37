"""Data-parallel runtime: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) — the replacement of tf.distribute.{OneDevice,Mirrored}Strategy selected by the reference's
get_strategy (retinanet/distribute.py:7-60).

Collectives of one training step and where they are issued here:
  C1  gradient all-reduce SUM after local clipping (executor.py:432-437)  -> all_reduce_sum_bucketed
  C2  loss normaliser all-reduce / replicas (retinanet_loss.py:46-49)     -> global_normalizer
  C3  SyncBatchNorm statistics (model/utils.py:10-12)                     -> TrainEngine, per conv group
  C4  metric mean over replicas (executor.py:450-452)                     -> Strategy.reduce_mean
  C5  detection gather (executor.py:397-398)                              -> Strategy.gather
xGMI is point-to-point, so C1 goes out in a few large buckets (default 64 MiB) rather than per
tensor: the flat gradient arena makes a bucket a contiguous slice, no packing kernels.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class Strategy:
    def __init__(self, kind, device, rank=0, world=1, group=None):
        self.kind, self.device, self.rank, self.world, self.group = kind, device, rank, world, group

    @property
    def num_replicas_in_sync(self):
        return self.world

    def reduce_mean(self, t):
        if self.world > 1:
            t = t.clone()
            dist.all_reduce(t, group=self.group)
            t /= self.world
        return t

    def gather(self, t):
        if self.world == 1:
            return t
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t.contiguous(), group=self.group)
        return torch.cat(out, dim=0)

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)

    def any_true(self, flag):
        """collective OR of a per-rank condition (loop control that every rank must leave together)"""
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0)

    def broadcast_object(self, obj, src=0):
        if self.world == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]


def get_strategy(params):
    """params = config `training.strategy` ({'type': 'gpu'|'cpu'|'multi_gpu'|'tpu', 'name': ...})."""
    kind = params.type
    if kind == "tpu":
        raise ValueError("Unsupported strategy requested: this build targets MI355X (use 'gpu' or 'multi_gpu')")
    if kind == "cpu":
        raise ValueError("strategy 'cpu' is not served: the product has no CPU path (see oracle/ for the CPU restatement)")
    if kind == "gpu":
        return Strategy("gpu", torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    if kind == "multi_gpu":
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        dev = torch.device("cuda", local)
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        return Strategy("multi_gpu", dev, rank, world)
    raise ValueError("Unsupported strategy requested")


def global_normalizer(num_positives_sum, world, group=None):
    """retinanet_loss.py:38-49: all_reduce_sum(sum(num-positives) + 1) / replicas, f32[1]."""
    n = (num_positives_sum + 1.0).reshape(1).to(torch.float32)
    if world > 1:
        dist.all_reduce(n, group=group)
        n = n / world
    return n


def syncbn_merge(sums, world, all_reduce, c2_local=None):
    """C3 (+ C2): the SyncBatchNorm statistics message of one BatchNorm group (model/utils.py:10-12: TF's
    SyncBatchNormalization all-reduces the per-replica sums).  `sums` = f32 [sum x | sum x^2 per channel of every layer of the
    group ..., one spare slot]: SUM-all-reduced in place through `all_reduce(tensor)`; the caller finalises with count x
    `world`.  When `c2_local` (this replica's sum(num-positives) + 1, f32[1]) is given it rides in the spare slot of this
    message — the step's first — instead of being a collective of its own (retinanet_loss.py:46-49), and the function
    returns the loss normaliser all_reduce_sum(...) / replicas as a view of the message; otherwise None.
    One function for TrainEngine._bn_stats_finalize and the world-size-2 CPU test."""
    if c2_local is not None:
        sums[-1:].copy_(c2_local)
    all_reduce(sums)
    return sums[-1:] / float(world) if c2_local is not None else None


def all_reduce_sum_bucketed(flat, world, group=None, bucket_bytes=64 << 20):
    """In-place SUM all-reduce of a flat tensor in contiguous buckets (async, then wait)."""
    if world <= 1:
        return flat
    step = max(1, bucket_bytes // flat.element_size())
    works = []
    for off in range(0, flat.numel(), step):
        works.append(dist.all_reduce(flat[off:off + step], group=group, async_op=True))
    for w in works:
        w.wait()
    return flat
