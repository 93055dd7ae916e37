"""TFRecord reading and `parse_example` (reference retinanet/dataloader/tfrecord_parser.py:4-41 and the
`tf.data.TFRecordDataset` it is mapped over, input_pipeline.py:60-68) — SURVEY §8(f)-4.

The record framing / CRC-32C check and the tf.train.Example wire format are parsed natively
(`rn_tfrecord_scan`, `rn_example_parse` in csrc/rn_tfrecord.hip); this module is the thin host mirror:

    for payload in TFRecordDataset(files):      # bytes of one serialized Example
        sample = parse_example(payload)         # {'image', 'image_id', 'objects': {'bbox', 'label'}}

`sample['image']` is float32 [h, w, 3] like `tf.cast(tf.io.decode_image(..., channels=3), tf.float32)`.
PNG and BMP are decoded here (zlib + numpy); baseline JPEG by the native decoder (`rn_jpeg_decode`, csrc/rn_jpeg.hip:
bit-identical to libjpeg-turbo); progressive / CMYK / arithmetic-coded JPEG fall back to Pillow when it is installed
(logged once), else raise `ImageDecodeError`.
`parse_example(..., decode=False)` returns the encoded bytes instead.
"""
from __future__ import annotations

import ctypes
import mmap
import os
import struct
import zlib

import numpy as np

from retinanet import _C


class ImageDecodeError(ValueError):
    pass


class DataLossError(IOError):
    """Corrupted / truncated TFRecord (tf.errors.DataLossError)."""


# ---------------------------------------------------------------------------------------------------
# record reader
class TFRecordDataset:
    """Iterates the payloads of one or more TFRecord files in order (tf.data.TFRecordDataset semantics:
    files one after the other, every record checked against both masked CRC-32Cs)."""

    def __init__(self, filenames, buffer_size=None, verify_crc=True):
        self.filenames = [filenames] if isinstance(filenames, (str, os.PathLike)) else list(filenames)
        self.verify_crc = bool(verify_crc)   # buffer_size is accepted for signature parity; files are mmap'd

    @staticmethod
    def index_file(path, verify_crc=True):
        """(mmap, offsets u64[n], lengths u64[n]) of one file."""
        lib = _C.lib()
        size = os.path.getsize(path)
        if size == 0:
            return None, np.zeros((0,), np.uint64), np.zeros((0,), np.uint64)
        with open(path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        buf = np.frombuffer(mm, dtype=np.uint8)
        cap = max(16, size // 64)
        while True:
            offs = np.empty((cap,), np.uint64)
            lens = np.empty((cap,), np.uint64)
            consumed = ctypes.c_size_t(0)
            n = lib.rn_tfrecord_scan(buf.ctypes.data, size, offs.ctypes.data, lens.ctypes.data, cap,
                                     1 if verify_crc else 0, 0, ctypes.byref(consumed))
            if n < 0:
                raise DataLossError(f"{path}: {lib.rn_last_error().decode()}")
            if consumed.value == size:
                return mm, offs[:n], lens[:n]
            cap *= 4   # more records than the first guess

    def __iter__(self):
        for path in self.filenames:
            mm, offs, lens = self.index_file(path, self.verify_crc)
            for o, l in zip(offs.tolist(), lens.tolist()):
                yield mm[o:o + l]


# ---------------------------------------------------------------------------------------------------
# image decoding (tf.io.decode_image(channels=3) for the formats that need no third-party codec)
def _png_decode(data):
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ImageDecodeError("not a PNG")
    pos, idat, ihdr, plte, trns = 8, [], None, None, None
    while pos + 8 <= len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    if ihdr is None:
        raise ImageDecodeError("PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = ihdr
    if interlace or (ctype == 3 and depth != 8) or (ctype != 3 and depth not in (8, 16)):
        raise ImageDecodeError(f"unsupported PNG variant (depth {depth}, colour type {ctype}, interlace {interlace})")
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bps = depth // 8
    bpp = nch * bps
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    if raw.size != h * (stride + 1):
        raise ImageDecodeError("PNG data size mismatch")
    rows = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros((stride,), np.int32)
    for y in range(h):
        ft = int(rows[y, 0])
        line = rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:   # Sub: prefix sums per byte lane
            cur = line.reshape(-1, bpp).cumsum(axis=0).reshape(-1) & 255
        else:           # Average / Paeth: sequential in x
            cur = np.zeros((stride,), np.int32)
            ln, pv = line.tolist(), prev.tolist()
            c = [0] * stride
            for i in range(stride):
                a = c[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ft == 3:
                    c[i] = (ln[i] + ((a + b) >> 1)) & 255
                else:
                    cc = pv[i - bpp] if i >= bpp else 0
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                    c[i] = (ln[i] + pr) & 255
            cur = np.asarray(c, np.int32)
        out[y] = cur
        prev = cur
    px = out.reshape(h, w, nch, bps)[..., 0]       # 16-bit: decode_image(dtype=uint8) keeps the high byte
    if ctype == 3:
        if plte is None:
            raise ImageDecodeError("palette PNG without PLTE")
        return plte[px[..., 0]]
    if ctype in (0, 4):
        return np.repeat(px[..., :1], 3, axis=2)
    return px[..., :3]


def _bmp_decode(data):
    if data[:2] != b"BM":
        raise ImageDecodeError("not a BMP")
    off = struct.unpack("<I", data[10:14])[0]
    w, h, _, bpp, comp = struct.unpack("<iiHHI", data[18:34])
    if comp != 0 or bpp not in (24, 32):
        raise ImageDecodeError("unsupported BMP variant")
    top_down = h < 0
    h = abs(h)
    nch = bpp // 8
    stride = (w * nch + 3) & ~3
    rows = np.frombuffer(data, np.uint8, count=stride * h, offset=off).reshape(h, stride)[:, :w * nch]
    img = rows.reshape(h, w, nch)[..., 2::-1]        # BGR(A) -> RGB
    return img if top_down else img[::-1]


_PILLOW_WARNED = False


def _jpeg_decode_pillow(data, why):
    """JPEG flavours the native baseline decoder refuses (progressive, CMYK / Adobe, arithmetic coding — COCO holds a few
    progressive files; tf.io.decode_image takes them all): Pillow's libjpeg-turbo, when it is installed, with ONE logged
    warning per process.  Pillow's default decode parameters are libjpeg's (islow IDCT, fancy up-sampling), the ones the
    native decoder restates."""
    global _PILLOW_WARNED
    try:
        import io

        from PIL import Image
    except ImportError:
        raise ImageDecodeError(why + " (and Pillow is not installed for the fallback)")
    if not _PILLOW_WARNED:
        import logging
        logging.warning("JPEG not decodable by the native baseline decoder (%s): falling back to Pillow for such records", why)
        _PILLOW_WARNED = True
    try:
        with Image.open(io.BytesIO(data)) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))
    except Exception as e:   # noqa: BLE001 — corrupt bytes: the pipeline's own error type
        raise ImageDecodeError(f"{why}; Pillow: {e}")


def _jpeg_decode(data):
    """baseline JPEG -> uint8 [h, w, 3] through the native decoder (csrc/rn_jpeg.hip: libjpeg's islow IDCT, fancy
    chroma up-sampling and YCbCr tables restated — bit-identical to libjpeg-turbo, tests/test_jpeg_cpu.py; grayscale
    replicated like decode_image(channels=3)); files it does not support go to Pillow."""
    lib = _C.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    w, h, c = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    # RN_EUNSUPPORTED (a status code, not message text): a well-formed file of a kind the native decoder does not
    # implement goes to Pillow; anything else non-zero is corrupt input
    rc = lib.rn_jpeg_info(buf.ctypes.data, buf.size, ctypes.byref(w), ctypes.byref(h), ctypes.byref(c))
    if rc != 0:
        why = lib.rn_last_error().decode()
        if rc == _C.RN_EUNSUPPORTED:
            return _jpeg_decode_pillow(data, why)
        raise ImageDecodeError(why)
    out = np.empty((h.value, w.value, 3), np.uint8)
    rc = lib.rn_jpeg_decode(buf.ctypes.data, buf.size, out.ctypes.data, out.size)
    if rc != 0:
        why = lib.rn_last_error().decode()
        if rc == _C.RN_EUNSUPPORTED:
            return _jpeg_decode_pillow(data, why)
        raise ImageDecodeError(why)
    return out


def decode_image(data, channels=3):
    """uint8 [h, w, 3]."""
    if channels != 3:
        raise ValueError("only channels=3 is used by the reference")
    data = bytes(data)
    if data[:8] == b"\x89PNG\r\n\x1a\n":
        return np.ascontiguousarray(_png_decode(data))
    if data[:2] == b"BM":
        return np.ascontiguousarray(_bmp_decode(data))
    if data[:3] == b"\xff\xd8\xff":
        return _jpeg_decode(data)
    if data[:6] in (b"GIF87a", b"GIF89a"):
        raise ImageDecodeError("GIF records are not decoded by this build (no detection data set ships them)")
    raise ImageDecodeError("unknown image format (tf.io.decode_image: BMP, GIF, JPEG or PNG)")


# ---------------------------------------------------------------------------------------------------
def parse_example(example_proto, decode=True):
    """tfrecord_parser.py:4-41.  `example_proto`: bytes / memoryview of one serialized tf.train.Example."""
    lib = _C.lib()
    rec = np.frombuffer(example_proto, dtype=np.uint8)
    info = _C.ExampleInfo()
    st = lib.rn_example_parse(rec.ctypes.data, rec.size, ctypes.byref(info), None, None, None, None, None, 0)
    if st != 0:
        raise ValueError(lib.rn_last_error().decode())
    cap = max(info.n_xmins, info.n_ymins, info.n_xmaxs, info.n_ymaxs, info.n_classes, 1)
    cols = np.zeros((4, cap), np.float32)
    classes = np.zeros((cap,), np.int64)
    st = lib.rn_example_parse(rec.ctypes.data, rec.size, ctypes.byref(info), cols[0].ctypes.data, cols[1].ctypes.data,
                              cols[2].ctypes.data, cols[3].ctypes.data, classes.ctypes.data, cap)
    if st != 0:
        raise ValueError(lib.rn_last_error().decode())
    n = (info.n_xmins, info.n_ymins, info.n_xmaxs, info.n_ymaxs)
    if len(set(n)) != 1:   # tf.stack of unequal sparse.to_dense vectors fails in the reference too
        raise ValueError(f"xmins/ymins/xmaxs/ymaxs have different lengths: {n}")
    enc = rec[info.image_offset:info.image_offset + info.image_length]
    image = decode_image(enc.tobytes()).astype(np.float32) if decode else enc.tobytes()
    return {"image": image, "image_id": int(info.image_id),
            "objects": {"bbox": np.ascontiguousarray(cols[:, :n[0]].T), "label": classes[:info.n_classes].copy()}}
