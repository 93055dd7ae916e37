"""CPU: SURVEY §8(f)-4 — TFRecord framing, tf.train.Example wire format, writer sharding and the input
pipeline's file / interleave / shuffle policy (reference dataloader/tfrecord_parser.py:4-41,
dataset_utils/tfrecord_writer.py:7-80, dataloader/input_pipeline.py:27-92).

Pins: CRC-32C against the RFC 3720 (iSCSI) B.4 test vectors; the Example parser / serialiser against the official
protobuf runtime (google.protobuf, with the tf.train.Example schema declared from its published .proto); the PNG
decoder against images filtered with every PNG filter type.  Host functions only: no GPU call."""
import ctypes
import os
import struct
import zlib

import numpy as np
import pytest

from retinanet import _C
from retinanet.dataloader.input_pipeline import InputContext, InputPipeline, interleave, shuffle_buffer
from retinanet.dataloader.tfrecord_parser import (DataLossError, ImageDecodeError, TFRecordDataset, decode_image,
                                                  parse_example)
from retinanet.dataset_utils.tfrecord_writer import TFrecordWriter, frame_record, serialize_example


# ---- the official protobuf runtime as the independent reader / writer ---------------------------------
def _example_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name="rnet_test_example.proto", package="rnet_test", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = f.message_type.add()
        m.name = name
        return m

    def field(m, name, num, typ, label=T.LABEL_OPTIONAL, type_name=None, oneof=None):
        fd = m.field.add(name=name, number=num, type=typ, label=label)
        if type_name:
            fd.type_name = type_name
        if oneof is not None:
            fd.oneof_index = oneof
        return fd

    field(msg("BytesList"), "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    field(msg("FloatList"), "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED)
    field(msg("Int64List"), "value", 1, T.TYPE_INT64, T.LABEL_REPEATED)
    feat = msg("Feature")
    feat.oneof_decl.add(name="kind")
    field(feat, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".rnet_test.BytesList", oneof=0)
    field(feat, "float_list", 2, T.TYPE_MESSAGE, type_name=".rnet_test.FloatList", oneof=0)
    field(feat, "int64_list", 3, T.TYPE_MESSAGE, type_name=".rnet_test.Int64List", oneof=0)
    feats = msg("Features")
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    field(entry, "key", 1, T.TYPE_STRING)
    field(entry, "value", 2, T.TYPE_MESSAGE, type_name=".rnet_test.Feature")
    field(feats, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".rnet_test.Features.FeatureEntry")
    field(msg("Example"), "features", 1, T.TYPE_MESSAGE, type_name=".rnet_test.Features")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("rnet_test.Example"))


@pytest.fixture(scope="module")
def Example():
    return _example_classes()


def _pb_example(Example, image, boxes, classes, image_id):
    ex = Example()
    fm = ex.features.feature
    fm["image"].bytes_list.value.append(image)
    fm["image_id"].int64_list.value.append(image_id)
    for k, col in (("xmins", 0), ("ymins", 1), ("xmaxs", 2), ("ymaxs", 3)):
        fm[k].float_list.value.extend([float(v) for v in boxes[:, col]])
    fm["classes"].int64_list.value.extend([int(c) for c in classes])
    return ex


# ---- CRC-32C ---------------------------------------------------------------------------------------------
def _crc(b):
    a = np.frombuffer(bytes(b), np.uint8)
    return _C.lib().rn_crc32c(a.ctypes.data if a.size else None, a.size)


def test_crc32c_rfc3720_vectors():
    assert _crc(b"\x00" * 32) == 0x8A9136AA
    assert _crc(b"\xff" * 32) == 0x62A8AB43
    assert _crc(bytes(range(32))) == 0x46DD794E
    assert _crc(bytes(range(31, -1, -1))) == 0x113FDB5C
    assert _crc(b"123456789") == 0xE3069283
    assert _crc(b"") == 0
    # unaligned starts and every tail length go through the byte loops around the 8-byte slices
    data = bytes((i * 131 + 7) & 255 for i in range(200))
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        table.append(c)
    for start in range(9):
        for n in (0, 1, 7, 8, 9, 63, 64, 65, 190):
            c = 0xFFFFFFFF
            for b in data[start:start + n]:
                c = (c >> 8) ^ table[(c ^ b) & 255]
            buf = np.frombuffer(data, np.uint8)
            assert _C.lib().rn_crc32c(buf.ctypes.data + start, n) == (c ^ 0xFFFFFFFF)


def test_masked_crc_and_frame_layout():
    payload = b"hello tfrecord"
    rec = frame_record(payload)
    assert len(rec) == len(payload) + 16
    (n,) = struct.unpack("<Q", rec[:8])
    assert n == len(payload) and rec[12:12 + n] == payload

    def mask(c):
        return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF

    assert struct.unpack("<I", rec[8:12])[0] == mask(_crc(rec[:8]))
    assert struct.unpack("<I", rec[-4:])[0] == mask(_crc(payload))


def test_scan_roundtrip_corruption_and_truncation(tmp_path):
    payloads = [b"", b"a", os.urandom(1000), b"x" * 70000]
    blob = b"".join(frame_record(p) for p in payloads)
    path = tmp_path / "a.tfrecord"
    path.write_bytes(blob)
    assert [bytes(p) for p in TFRecordDataset(str(path))] == payloads
    # flipped payload byte -> data CRC mismatch; flipped length byte -> length CRC mismatch; cut tail -> truncated
    for pos, what in ((12 + 16 + 1 + 16 + 5, "corrupted record data"), (16 + 2, "corrupted record length")):
        bad = bytearray(blob)
        bad[pos] ^= 0x40
        path.write_bytes(bytes(bad))
        with pytest.raises(DataLossError, match=what):
            list(TFRecordDataset(str(path)))
    path.write_bytes(blob[:-3])
    with pytest.raises(DataLossError, match="truncated"):
        list(TFRecordDataset(str(path)))
    # a streaming reader may hand over a partial tail: it is left unconsumed
    lib = _C.lib()
    buf = np.frombuffer(blob[:-3], np.uint8)
    offs, lens = np.zeros(8, np.uint64), np.zeros(8, np.uint64)
    consumed = ctypes.c_size_t(0)
    n = lib.rn_tfrecord_scan(buf.ctypes.data, buf.size, offs.ctypes.data, lens.ctypes.data, 8, 1, 1, ctypes.byref(consumed))
    assert n == 3 and consumed.value == sum(len(p) + 16 for p in payloads[:3])
    empty = tmp_path / "empty.tfrecord"
    empty.write_bytes(b"")
    assert list(TFRecordDataset(str(empty))) == []


# ---- Example wire format --------------------------------------------------------------------------------
def _rand_sample(rng, n):
    boxes = rng.uniform(0, 1, size=(n, 4)).astype(np.float32)
    classes = rng.integers(0, 80, size=(n,)).astype(np.int64)
    image = bytes(rng.integers(0, 256, size=int(rng.integers(1, 500))).astype(np.uint8))
    return image, boxes, classes, int(rng.integers(0, 2 ** 40))


@pytest.mark.parametrize("n", [0, 1, 7, 200])
def test_parse_matches_official_protobuf_writer(Example, n):
    rng = np.random.default_rng(n)
    image, boxes, classes, image_id = _rand_sample(rng, n)
    if n == 7:
        classes[0], image_id = -3, -5          # negative int64: 10-byte varints
    data = _pb_example(Example, image, boxes, classes, image_id).SerializeToString()
    s = parse_example(data, decode=False)
    assert s["image"] == image and s["image_id"] == image_id
    np.testing.assert_array_equal(s["objects"]["bbox"], boxes)
    np.testing.assert_array_equal(s["objects"]["label"], classes)
    assert s["objects"]["bbox"].shape == (n, 4) and s["objects"]["bbox"].dtype == np.float32


@pytest.mark.parametrize("n", [0, 1, 33])
def test_serialize_is_read_by_official_protobuf(Example, n):
    rng = np.random.default_rng(100 + n)
    image, boxes, classes, image_id = _rand_sample(rng, n)
    data = serialize_example(image, boxes, classes, image_id)
    ex = Example()
    ex.ParseFromString(data)
    fm = ex.features.feature
    assert sorted(fm.keys()) == ["classes", "image", "image_id", "xmaxs", "xmins", "ymaxs", "ymins"]
    assert list(fm["image"].bytes_list.value) == [image]
    assert list(fm["image_id"].int64_list.value) == [image_id]
    for k, col in (("xmins", 0), ("ymins", 1), ("xmaxs", 2), ("ymaxs", 3)):
        np.testing.assert_array_equal(np.float32(list(fm[k].float_list.value)), boxes[:, col])
    assert list(fm["classes"].int64_list.value) == classes.tolist()
    # and our own reader takes it back
    s = parse_example(data, decode=False)
    np.testing.assert_array_equal(s["objects"]["bbox"], boxes)
    assert s["image_id"] == image_id
    # same size as the official runtime's own serialisation of that message (map order is unspecified)
    assert len(ex.SerializeToString()) == len(data)


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _ld(field, body):
    return _varint((field << 3) | 2) + _varint(len(body)) + body


def _entry(key, kind, list_body):
    return _ld(1, _ld(1, key.encode()) + _ld(2, _ld(kind, list_body)))


def test_unpacked_lists_unknown_fields_and_duplicate_keys():
    """proto2-style writers emit repeated scalars unpacked; unknown keys / fields are skipped; for a repeated map
    key the last entry wins."""
    floats = [0.25, 0.5]
    unpacked_f = b"".join(_varint((1 << 3) | 5) + struct.pack("<f", v) for v in floats)
    unpacked_i = b"".join(_varint((1 << 3) | 0) + _varint(v) for v in (3, 70))
    feats = (_entry("xmins", 2, unpacked_f) + _entry("ymins", 2, unpacked_f) + _entry("xmaxs", 2, unpacked_f) +
             _entry("ymaxs", 2, unpacked_f) + _entry("classes", 3, unpacked_i) +
             _entry("something_else", 1, _ld(1, b"zz")) +
             _entry("image_id", 3, _ld(1, _varint(1))) + _entry("image_id", 3, _ld(1, _varint(42))) +
             _entry("image", 1, _ld(1, b"PIXELS")) + _varint((9 << 3) | 0) + _varint(5))
    data = _ld(1, feats) + _varint((7 << 3) | 5) + b"\x00\x00\x00\x00"
    s = parse_example(data, decode=False)
    assert s["image"] == b"PIXELS" and s["image_id"] == 42
    np.testing.assert_array_equal(s["objects"]["bbox"], np.float32([[0.25] * 4, [0.5] * 4]))
    assert s["objects"]["label"].tolist() == [3, 70]


def test_parse_errors_like_parse_single_example(Example):
    rng = np.random.default_rng(5)
    image, boxes, classes, image_id = _rand_sample(rng, 3)
    ex = _pb_example(Example, image, boxes, classes, image_id)
    del ex.features.feature["image"]
    with pytest.raises(ValueError, match="Feature: image .data type: string. is required"):
        parse_example(ex.SerializeToString(), decode=False)
    ex = _pb_example(Example, image, boxes, classes, image_id)
    ex.features.feature["image_id"].float_list.value.append(1.0)      # oneof switches to float_list
    with pytest.raises(ValueError, match="image_id.*Data types don't match"):
        parse_example(ex.SerializeToString(), decode=False)
    ex = _pb_example(Example, image, boxes, classes, image_id)
    ex.features.feature["image_id"].int64_list.value.append(2)
    with pytest.raises(ValueError, match="image_id"):
        parse_example(ex.SerializeToString(), decode=False)
    ex = _pb_example(Example, image, boxes, classes, image_id)
    ex.features.feature["xmins"].float_list.value.append(0.5)         # ragged box columns: tf.stack fails too
    with pytest.raises(ValueError, match="different lengths"):
        parse_example(ex.SerializeToString(), decode=False)
    good = _pb_example(Example, image, boxes, classes, image_id).SerializeToString()
    for cut in (1, len(good) // 2, len(good) - 1):
        with pytest.raises(ValueError):
            parse_example(good[:cut], decode=False)
    # VarLen features may be absent altogether
    ex = Example()
    ex.features.feature["image"].bytes_list.value.append(b"i")
    ex.features.feature["image_id"].int64_list.value.append(9)
    s = parse_example(ex.SerializeToString(), decode=False)
    assert s["objects"]["bbox"].shape == (0, 4) and s["objects"]["label"].shape == (0,)


# ---- image decoding ---------------------------------------------------------------------------------------
def _png_encode(img, filters):
    """Minimal PNG writer that applies the given filter type per row (cycled)."""
    h, w, c = img.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    bpp = c
    raw = bytearray()
    prev = np.zeros((w * c,), np.int32)
    for y in range(h):
        line = img[y].reshape(-1).astype(np.int32)
        ft = filters[y % len(filters)]
        a = np.concatenate([np.zeros(bpp, np.int32), line[:-bpp]])
        cc = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = line
        elif ft == 1:
            f = line - a
        elif ft == 2:
            f = line - prev
        elif ft == 3:
            f = line - ((a + prev) >> 1)
        else:
            p = a + prev - cc
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - cc)
            pr = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, cc))
            f = line - pr
        raw.append(ft)
        raw += bytes((f & 255).astype(np.uint8))
        prev = line

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(bytes(raw))[:40]) + chunk(b"IDAT", zlib.compress(bytes(raw))[40:]) +
            chunk(b"IEND", b""))


@pytest.mark.parametrize("channels", [1, 3, 4])
def test_png_decode_all_filter_types(channels):
    rng = np.random.default_rng(channels)
    img = rng.integers(0, 256, size=(11, 13, channels)).astype(np.uint8)
    got = decode_image(_png_encode(img, [0, 1, 2, 3, 4]))
    want = np.repeat(img, 3, axis=2) if channels == 1 else img[..., :3]
    np.testing.assert_array_equal(got, want)
    assert got.dtype == np.uint8 and got.shape == (11, 13, 3)


def test_bmp_decode_and_unknown_format():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(5, 7, 3)).astype(np.uint8)
    stride = (7 * 3 + 3) & ~3
    rows = b"".join(bytes(img[y, :, ::-1].reshape(-1)) + b"\x00" * (stride - 21) for y in range(4, -1, -1))
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 7, 5, 1, 24, 0,
                                                                               len(rows), 0, 0, 0, 0)
    np.testing.assert_array_equal(decode_image(hdr + rows), img)
    with pytest.raises(ImageDecodeError):
        decode_image(b"not an image at all")


# ---- writer sharding + pipeline policy --------------------------------------------------------------------
def _write_dataset(tmp_path, n_samples, n_shards, prefix="train", size=(9, 12)):
    rng = np.random.default_rng(1)
    w = TFrecordWriter(n_samples, n_shards, output_dir=str(tmp_path), prefix=prefix)
    for i in range(n_samples):
        img = rng.integers(0, 256, size=(size[0], size[1], 3)).astype(np.uint8)
        n = int(rng.integers(0, 4))
        lo = rng.uniform(0, 0.5, size=(n, 2))
        boxes = np.concatenate([lo, lo + rng.uniform(0.1, 0.5, size=(n, 2))], axis=1).astype(np.float32)
        w.push(_png_encode(img, [i % 5]), boxes, rng.integers(0, 80, size=(n,)), 1000 + i)
    w.flush_last()
    return sorted(str(p) for p in tmp_path.glob(prefix + "-*.tfrecord"))


def test_writer_shards_like_the_reference(tmp_path):
    files = _write_dataset(tmp_path, 10, 3)
    assert [os.path.basename(f) for f in files] == ["train-0001.tfrecord", "train-0002.tfrecord", "train-0003.tfrecord"]
    counts = [len(list(TFRecordDataset(f))) for f in files]
    assert counts == [3, 3, 4]        # step 10 // 3, the remainder goes to the last shard (:14-15, :62-64)
    ids = [parse_example(r)["image_id"] for f in files for r in TFRecordDataset(f)]
    assert ids == list(range(1000, 1010))
    s = parse_example(next(iter(TFRecordDataset(files[0]))))
    assert s["image"].dtype == np.float32 and s["image"].shape == (9, 12, 3)


def test_interleave_and_shuffle_buffer():
    # expected orders traced by hand through tf.data's InterleaveDataset iterator: an exhausted slot is refilled
    # (and yields) when the cycle comes back to it
    srcs = [[f"{c}{i}" for i in range(n)] for c, n in (("a", 3), ("b", 1), ("c", 4), ("d", 2))]
    assert list(interleave(srcs, 2)) == ["a0", "b0", "a1", "a2", "c0", "c1", "d0", "c2", "d1", "c3"]
    assert list(interleave(srcs, 8)) == ["a0", "b0", "c0", "d0", "a1", "c1", "d1", "a2", "c2", "c3"]
    assert list(interleave(srcs, 2, block_length=2)) == ["a0", "a1", "b0", "a2", "c0", "c1", "d0", "d1", "c2", "c3"]
    out = list(shuffle_buffer(iter(range(100)), 16, np.random.default_rng(0)))
    assert sorted(out) == list(range(100)) and out != list(range(100))
    assert all(v < 16 + i + 1 for i, v in enumerate(out))   # an element cannot be emitted before it was buffered
    assert list(shuffle_buffer(iter(range(5)), 1, np.random.default_rng(0))) == list(range(5))


def test_pipeline_file_policy(tmp_path, params):
    import copy
    files = _write_dataset(tmp_path, 16, 8)
    p = copy.deepcopy(params)
    p.dataloader_params["tfrecords"] = {"train": str(tmp_path / "train-*"), "val": str(tmp_path / "train-*")}
    p.dataloader_params["shuffle_buffer_size"] = 4
    p.dataloader_params["augmentations"] = {"use_augmentation": False}
    with pytest.raises(AssertionError):
        InputPipeline("test", p, False, 1)
    val = InputPipeline("val", p, True, 2)
    seen = []
    for rank in range(2):
        mine = list(val._files(InputContext(2, rank, 2)))
        assert len(mine) == 4
        seen += mine
    assert sorted(seen) == files                         # shards are disjoint and complete
    assert list(val._files(None)) != files               # shuffled ...
    assert list(val._files(None)) == list(val._files(None))   # ... with the fixed seed
    train = InputPipeline("train", p, False, 1)
    it = train._files(None)
    first, second = [next(it) for _ in range(8)], [next(it) for _ in range(8)]
    assert sorted(first) == files and sorted(second) == files and first != second   # repeat + reshuffle
    assert InputContext(1, 0, 8).get_per_replica_batch_size(256) == 32
    # records of a val pass: every sample exactly once
    val.cycle_length = 3
    ids = sorted(parse_example(r, decode=False)["image_id"] for r in val._records(None))
    assert ids == list(range(1000, 1016))


def test_golden_tfrecord_written_by_independent_tools(tmp_path):
    """tests/golden/example_golden.npz (make_example_golden.py): Examples serialised by the official protobuf runtime,
    framed with a bit-by-bit CRC-32C — read back through TFRecordDataset + parse_example."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "example_golden.npz"))
    path = tmp_path / "golden.tfrecord"
    path.write_bytes(z["tfrecord"].tobytes())
    recs = list(TFRecordDataset(str(path)))
    assert len(recs) == 4
    io, bo = 0, 0
    for i, rec in enumerate(recs):
        s = parse_example(rec)
        h, w = z["hw"][i]
        n = int(z["counts"][i])
        np.testing.assert_array_equal(s["image"], z["images"][io:io + h * w * 3].reshape(h, w, 3).astype(np.float32))
        np.testing.assert_array_equal(s["objects"]["bbox"], z["boxes"][bo * 4:(bo + n) * 4].reshape(n, 4))
        np.testing.assert_array_equal(s["objects"]["label"], z["classes"][bo:bo + n])
        assert s["image_id"] == int(z["image_ids"][i])
        io += h * w * 3
        bo += n
