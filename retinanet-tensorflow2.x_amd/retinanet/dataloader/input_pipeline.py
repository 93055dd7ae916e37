"""`InputPipeline` — the reference's tf.data graph as a plain Python generator feeding the GPU
(retinanet/dataloader/input_pipeline.py:8-92) — SURVEY §8(f)-4.

Same stages in the same order:
  glob (sorted, like tf.io.gfile.glob) -> file shuffle (seed 1337, reshuffled every epoch) -> shard by input
  pipeline (`files[id::n]`) -> repeat (train) -> deterministic interleave
  (cycle_length = number of CPU cores as AUTOTUNE picks, block_length 1) of TFRecordDataset -> [train] shuffle buffer
  -> map(parse_example + preprocessing) -> batch (train: drop_remainder; val: keep) -> encode.
What differs by design: the label encoding is not mapped per sample on the host but done once per batch on the GPU
(`LabelEncoder.encode_batch`: two HBM-bound launches for the whole batch), so a train batch is
`(images f32[B,H,W,3] cuda, targets)` with `targets` = the reference's dict per level with a leading batch axis;
random draws come from numpy Generators seeded with `_RANDOM_SEED` (TensorFlow's streams are not reproducible here).

Sharding: the reference shards files and divides the batch only when `is_multi_host` (:44-47) because on ONE host a
single process drives every GPU and MirroredStrategy splits the global batch.  This build runs one PROCESS per GPU, so
every rank is an input pipeline of its own: whenever the `input_context` says there is more than one pipeline the
files are sharded (`files[id::n]`, the FILE shuffle keeps the common seed so that the shards stay disjoint), the batch
size is the per-replica one, and the shuffle-buffer / augmentation seeds are offset by the pipeline id — with or
without `is_multi_host`.
"""
from __future__ import annotations

import glob as _glob
import logging
import os

import numpy as np
import torch

from retinanet.dataloader.label_encoder import LabelEncoder
from retinanet.dataloader.preprocessing_pipeline import PreprocessingPipeline
from retinanet.dataloader.tfrecord_parser import TFRecordDataset, parse_example


class InputContext:
    """The three fields of tf.distribute.InputContext the reference reads (:44-47)."""

    def __init__(self, num_input_pipelines=1, input_pipeline_id=0, num_replicas_in_sync=1):
        self.num_input_pipelines = num_input_pipelines
        self.input_pipeline_id = input_pipeline_id
        self.num_replicas_in_sync = num_replicas_in_sync

    def get_per_replica_batch_size(self, global_batch_size):
        if global_batch_size % self.num_replicas_in_sync:
            raise ValueError(f"The `global_batch_size` {global_batch_size} is not divisible by "
                             f"`num_replicas_in_sync` {self.num_replicas_in_sync}")
        return global_batch_size // self.num_replicas_in_sync


def interleave(sources, cycle_length, block_length=1):
    """tf.data interleave, deterministic: keep `cycle_length` iterators open, take `block_length` elements from
    each in turn, replace an exhausted one by the next source."""
    sources = iter(sources)
    slots = []
    for _ in range(cycle_length):
        nxt = next(sources, None)
        if nxt is None:
            break
        slots.append(iter(nxt))
    while slots:
        i = 0
        while i < len(slots):
            exhausted = False
            for _ in range(block_length):
                try:
                    yield next(slots[i])
                except StopIteration:
                    exhausted = True
                    break
            if exhausted:
                nxt = next(sources, None)
                if nxt is None:
                    slots.pop(i)
                    continue
                slots[i] = iter(nxt)
            i += 1


def shuffle_buffer(it, buffer_size, rng):
    """tf.data shuffle: fill a buffer, emit a uniformly chosen slot, refill it from the stream."""
    buf = []
    for x in it:
        if len(buf) < buffer_size:
            buf.append(x)
            continue
        j = int(rng.integers(0, buffer_size))
        yield buf[j]
        buf[j] = x
    while buf:
        j = int(rng.integers(0, len(buf)))
        buf[j], buf[-1] = buf[-1], buf[j]
        yield buf.pop()


class InputPipeline:
    _SUPPORTED_RUN_MODES = ["train", "val"]
    _RANDOM_SEED = 1337

    def __init__(self, run_mode, params, is_multi_host, num_replicas, device=None):
        if run_mode not in InputPipeline._SUPPORTED_RUN_MODES:
            raise AssertionError("Unsupported run mode requested, available run modes: {}".format(
                InputPipeline._SUPPORTED_RUN_MODES))
        self.run_mode = run_mode
        self.is_multi_host = is_multi_host
        self.num_replicas = num_replicas
        self.batch_size = params.training.batch_size[run_mode]
        self.shuffle_buffer_size = params.dataloader_params.shuffle_buffer_size
        self.tfrecord_files = params.dataloader_params.tfrecords[run_mode]
        self._params, self._device, self._label_encoder = params, device, None   # anchors are built on first use
        self.preprocessing_pipeline = PreprocessingPipeline(params.input.input_shape, params.dataloader_params)
        self.preprocessing_pipeline.rng = np.random.default_rng(InputPipeline._RANDOM_SEED + 2)
        self.cycle_length = os.cpu_count() or 1

    @property
    def label_encoder(self):
        if self._label_encoder is None:
            self._label_encoder = LabelEncoder(self._params, device=self._device)
        return self._label_encoder

    # -- stages -------------------------------------------------------------------------------------------
    def _files(self, input_context):
        matched = sorted(_glob.glob(self.tfrecord_files))
        logging.info("Found %d %s tfrecords matching %s", len(matched), self.run_mode, self.tfrecord_files)
        if not matched:
            raise FileNotFoundError(f"no tfrecords match {self.tfrecord_files}")
        rng = np.random.default_rng(InputPipeline._RANDOM_SEED)   # common to all ranks: disjoint shards of ONE order
        sharded = input_context is not None and input_context.num_input_pipelines > 1
        if sharded and len(matched) < input_context.num_input_pipelines:
            raise ValueError(f"{len(matched)} tfrecord files cannot be sharded over "
                             f"{input_context.num_input_pipelines} input pipelines")

        def epochs():
            while True:
                order = list(matched)
                rng.shuffle(order)          # dataset.shuffle(num_files, reshuffle_each_iteration=True)
                if sharded:                 # dataset.shard(num_input_pipelines, input_pipeline_id)
                    order = order[input_context.input_pipeline_id::input_context.num_input_pipelines]
                yield from order
                if self.run_mode != "train":
                    return

        return epochs()

    def _records(self, input_context):
        return interleave((TFRecordDataset(f) for f in self._files(input_context)), self.cycle_length, 1)

    def __call__(self, input_context=None):
        batch_size = self.batch_size
        pid = 0
        if input_context is not None and input_context.num_replicas_in_sync > 1:
            batch_size = input_context.get_per_replica_batch_size(self.batch_size)
            pid = int(input_context.input_pipeline_id)
        records = self._records(input_context)
        if self.run_mode == "val":
            return self._val_batches(records, batch_size)
        # per-pipeline streams: ranks must not draw the same shuffle order / flips / scale jitter
        self.preprocessing_pipeline.rng = np.random.default_rng([InputPipeline._RANDOM_SEED + 2, pid])
        records = shuffle_buffer(records, self.shuffle_buffer_size,
                                 np.random.default_rng([InputPipeline._RANDOM_SEED + 1, pid]))
        return self._train_batches(records, batch_size)

    def _val_batches(self, records, batch_size):
        batch = []
        for rec in records:
            batch.append(self.preprocessing_pipeline.preprocess_val_sample(_to_torch(parse_example(rec))))
            if len(batch) == batch_size:
                yield _stack_val(batch)
                batch = []
        if batch:                            # drop_remainder=False
            yield _stack_val(batch)

    def _train_batches(self, records, batch_size):
        images, boxes, classes = [], [], []
        for rec in records:
            img, b, c = self.preprocessing_pipeline(parse_example(rec))
            images.append(img)
            boxes.append(b)
            classes.append(c)
            if len(images) == batch_size:    # drop_remainder=True: a short tail is never emitted
                yield self._encode(images, boxes, classes)
                images, boxes, classes = [], [], []

    def _encode(self, images, boxes, classes):
        B = len(images)
        gmax = max(1, max(b.shape[0] for b in boxes))
        gt = np.zeros((B, gmax, 4), np.float32)
        cl = np.zeros((B, gmax), np.float32)
        cnt = np.zeros((B,), np.int32)
        for i, (b, c) in enumerate(zip(boxes, classes)):
            gt[i, :b.shape[0]] = b
            cl[i, :b.shape[0]] = c
            cnt[i] = b.shape[0]
        targets = self.label_encoder.encode_batch(torch.from_numpy(gt), torch.from_numpy(cl), torch.from_numpy(cnt))
        return torch.stack(images), targets


def _to_torch(sample):
    sample = dict(sample)
    sample["image"] = torch.from_numpy(sample["image"])
    return sample


def _stack_val(batch):
    return {"image": torch.stack([b["image"] for b in batch]),
            "image_id": torch.tensor([b["image_id"] for b in batch], dtype=torch.int64),
            "resize_scale": torch.stack([b["resize_scale"] for b in batch])}
