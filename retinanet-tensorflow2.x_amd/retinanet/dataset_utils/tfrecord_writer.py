"""`TFrecordWriter` with the reference's surface and sharding arithmetic
(retinanet/dataset_utils/tfrecord_writer.py:7-80) — SURVEY §8(f)-4.

`push(image_bytes, boxes[n,4] normalised (xmin,ymin,xmax,ymax), classes[n], image_id)` buffers samples and writes
`<prefix>-NNNN.tfrecord` every `n_samples // n_shards` samples (the last shard also takes the remainder);
`flush_last()` writes what is left.  Examples are serialised and framed natively (`rn_example_serialize`,
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
        self.n_samples = n_samples
        self.n_shards = n_shards
        self._step_size = self.n_samples // self.n_shards
        self.prefix = prefix
        self.output_dir = output_dir
        self._buffer = []
        self._file_count = 1
        self._remainder = self.n_samples - (self._step_size * self.n_shards)
        logging.info("writing %d samples in each tfrecord", self._step_size)
        if self._remainder:
            logging.warning("writing %d remaining samples in last tfrecord", self._remainder)

    @staticmethod
    def _make_example(image, boxes, classes, image_id):
        """Serialized tf.train.Example (the reference returns the message and serialises it in `_write_tfrecord`)."""
        return serialize_example(image, boxes, classes, image_id)

    def _write_tfrecord(self, tfrecord_path):
        if not self._buffer:
            logging.warning("no samples to be written")
            return
        logging.info("writing %d samples in %s", len(self._buffer), tfrecord_path)
        with open(tfrecord_path, "wb") as f:
            for (image, boxes, classes, image_id) in self._buffer:
                f.write(frame_record(TFrecordWriter._make_example(image, boxes, classes, image_id)))

    def _clear_buffer(self):
        self._buffer = []

    def _path(self):
        return os.path.join(self.output_dir, self.prefix + "-{:04.0f}".format(self._file_count) + ".tfrecord")

    def push(self, image, boxes, classes, image_id):
        self._buffer.append([image, boxes, classes, image_id])
        max_buffer_size = self._step_size
        if self._file_count == self.n_shards:
            max_buffer_size += self._remainder
        if len(self._buffer) == max_buffer_size:
            self._write_tfrecord(self._path())
            self._clear_buffer()
            self._file_count += 1

    def flush_last(self):
        if self._buffer:
            self._write_tfrecord(self._path())
