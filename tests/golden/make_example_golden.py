"""Generates tests/golden/example_golden.npz: tf.train.Example records serialised by the OFFICIAL protobuf runtime
(google.protobuf, schema declared from the published example.proto / feature.proto), framed as TFRecords with
CRC-32C computed by an independent bit-by-bit implementation in this script — i.e. nothing in the fixture was
produced by the library under test.  Run from the repo root:  python tests/golden/make_example_golden.py"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "retinanet-tensorflow2.x_amd"))
from test_tfrecord_cpu import _example_classes, _pb_example, _png_encode   # noqa: E402


def crc32c_bitwise(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def frame(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", mask(crc32c_bitwise(head))) + payload + struct.pack("<I", mask(crc32c_bitwise(payload)))


def main():
    Example = _example_classes()
    rng = np.random.default_rng(20240607)
    blob, images, boxes_all, classes_all, ids, hw = b"", [], [], [], [], []
    for i in range(4):
        h, w = int(rng.integers(3, 9)), int(rng.integers(3, 9))
        img = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
        n = [0, 1, 3, 6][i]
        boxes = rng.uniform(0, 1, size=(n, 4)).astype(np.float32)
        classes = rng.integers(0, 80, size=(n,)).astype(np.int64)
        image_id = [7, 2 ** 33 + 5, -1, 139][i]
        payload = _pb_example(Example, _png_encode(img, [i % 5, (i + 2) % 5]), boxes, classes, image_id).SerializeToString()
        blob += frame(payload)
        hw.append([h, w]); images.append(img.reshape(-1)); boxes_all.append(boxes.reshape(-1)); classes_all.append(classes); ids.append(image_id)
    np.savez(os.path.join(HERE, "example_golden.npz"), tfrecord=np.frombuffer(blob, np.uint8),
             images=np.concatenate(images), boxes=np.concatenate(boxes_all), classes=np.concatenate(classes_all),
             counts=np.asarray([len(c) for c in classes_all]), image_ids=np.asarray(ids, np.int64),
             hw=np.asarray(hw))
    print("wrote example_golden.npz,", len(blob), "bytes of TFRecord")


if __name__ == "__main__":
    main()
