"""`TFrecordWriter`: the reference's public surface (constructor arguments, `push`, `flush_last`, file names) and its shard
sizes (retinanet/dataset_utils/tfrecord_writer.py:7-80) — SURVEY §8(f)-4 — as a STREAMING writer.

`push(image_bytes, boxes[n,4] normalised (xmin,ymin,xmax,ymax), classes[n], image_id)` serialises and frames the sample
at once and appends it to the open shard `<prefix>-NNNN.tfrecord` (numbered from 1); shard i of `n_shards` closes after
`n_samples // n_shards` records, the last one after that plus the remainder; `flush_last()` closes a shard that is still
open (fewer samples pushed than announced).  Nothing but the current record is held in memory (the reference keeps a
whole shard of encoded images in a Python list before it writes).  Examples are serialised and framed natively (`rn_example_serialize`,
`rn_tfrecord_frame`): same features and dtypes as `_make_example` (:27-44), keys in sorted order, packed lists —
any protobuf reader (TensorFlow's included) parses them.
"""
from __future__ import annotations

import logging
import os

import numpy as np

from retinanet import _C


def serialize_example(image, boxes, classes, image_id):
    lib = _C.lib()
    img = np.frombuffer(bytes(image), dtype=np.uint8)
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.float32).reshape(-1, 4))
    classes = np.ascontiguousarray(np.asarray(classes, dtype=np.int64).reshape(-1))
    args = (img.ctypes.data if img.size else None, img.size, int(image_id), boxes.ctypes.data if boxes.size else None,
            boxes.shape[0], classes.ctypes.data if classes.size else None, classes.size)
    need = lib.rn_example_serialize(*args, None, 0)
    out = np.empty((need,), np.uint8)
    got = lib.rn_example_serialize(*args, out.ctypes.data, need)
    assert got == need
    return out.tobytes()


def frame_record(payload):
    lib = _C.lib()
    src = np.frombuffer(payload, dtype=np.uint8)
    out = np.empty((src.size + 16,), np.uint8)
    n = lib.rn_tfrecord_frame(src.ctypes.data if src.size else None, src.size, out.ctypes.data)
    return out[:n].tobytes()


class TFrecordWriter:
    def __init__(self, n_samples, n_shards, output_dir="", prefix=""):
        if n_shards < 1 or n_samples < n_shards:
            raise ValueError(f"cannot cut {n_samples} samples into {n_shards} shards")
        self.n_samples, self.n_shards = int(n_samples), int(n_shards)
        self.output_dir, self.prefix = output_dir, prefix
        per_shard, extra = divmod(self.n_samples, self.n_shards)
        # records each shard takes, in file order: the remainder rides in the last file (reference :14-15, :62-64)
        self._plan = [per_shard] * (self.n_shards - 1) + [per_shard + extra]
        self._shard = 0          # index into _plan of the shard being written
        self._in_shard = 0       # records already in it
        self._file = None
        logging.info("%d samples per tfrecord%s", per_shard, f", {extra} more in the last one" if extra else "")

    @staticmethod
    def _make_example(image, boxes, classes, image_id):
        """Serialized tf.train.Example with the features of the reference's `_make_example` (:27-44)."""
        return serialize_example(image, boxes, classes, image_id)

    def shard_path(self, index):
        """path of the index-th shard (0-based); the name carries the 1-based number in four digits"""
        return os.path.join(self.output_dir, f"{self.prefix}-{index + 1:04d}.tfrecord")

    def _close(self):
        if self._file is not None:
            self._file.close()
            logging.info("wrote %d samples to %s", self._in_shard, self._file.name)
            self._file = None
            self._shard += 1
            self._in_shard = 0

    def push(self, image, boxes, classes, image_id):
        if self._file is None:
            # samples past the announced total go to further files of the regular size, like the reference's counter
            self._file = open(self.shard_path(self._shard), "wb")
        self._file.write(frame_record(self._make_example(image, boxes, classes, image_id)))
        self._in_shard += 1
        quota = self._plan[self._shard] if self._shard < self.n_shards else self._plan[0]
        if self._in_shard == quota:
            self._close()

    def flush_last(self):
        if self._file is None:
            logging.warning("no samples to be written")
        self._close()
