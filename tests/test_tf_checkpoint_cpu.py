"""CPU: SURVEY §8(f)-3 — TensorFlow checkpoint files (tensor bundle = SSTable index + data shard, object-graph
string tensor) read and written without TensorFlow (reference: model.save_weights / load_weights,
executor.py:221-244, 652-654, 695-697; resnet.py:404-405).

PARITY UNPINNED against TensorFlow itself (not installable, no checkpoint fixture in the reference): these tests pin
the reader against hand-assembled bytes of the published formats (LevelDB block layout with prefix compression and
restarts, snappy elements, bundle / object-graph protos built with the official protobuf runtime) and the writer
against the reader."""
import os
import struct

import numpy as np
import pytest

from retinanet import tf_checkpoint as ck


def _mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _with_trailer(block, ctype=0):
    body = bytes(block) + bytes([ctype])
    return body + struct.pack("<I", _mask(ck._crc(body)))


def _hand_table(blocks):
    """blocks: list of (raw block bytes, compression type, last key) -> SSTable image assembled by hand."""
    out, handles = b"", []
    for blk, ctype, last in blocks:
        handles.append((last, ck._put_varint(len(out)) + ck._put_varint(len(blk))))
        out += _with_trailer(blk, ctype)
    meta = struct.pack("<II", 0, 1)
    mh = ck._put_varint(len(out)) + ck._put_varint(len(meta))
    out += _with_trailer(meta)
    idx = b""
    restarts = []
    for last, h in handles:                       # index block: restart interval 1, nothing shared
        restarts.append(len(idx))
        idx += ck._put_varint(0) + ck._put_varint(len(last)) + ck._put_varint(len(h)) + last + h
    idx += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    ih = ck._put_varint(len(out)) + ck._put_varint(len(idx))
    out += _with_trailer(idx)
    footer = mh + ih
    return out + footer + b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")


def test_table_reader_on_hand_built_blocks_prefix_compression_and_snappy():
    # block 1: three entries, the 2nd and 3rd share a prefix with their predecessor (LevelDB table_format.md)
    def ent(shared, tail, val):
        return ck._put_varint(shared) + ck._put_varint(len(tail)) + ck._put_varint(len(val)) + tail + val
    b1 = ent(0, b"apple", b"1") + ent(3, b"ly", b"22") + ent(5, b"x", b"") + struct.pack("<II", 0, 1)
    # block 2 is snappy compressed: literal "bananabana" expressed as literal "banana" + copy(offset 6, len 4),
    # wrapped in one entry  key "b", value "bananabana"
    raw2 = ent(0, b"b", b"bananabana") + struct.pack("<II", 0, 1)
    head = raw2[:3 + 1 + 6]                      # entry header + key + "banana"
    lit = bytes([(len(head) - 1) << 2]) + head
    copy = bytes([((4 - 4) << 2) | 1 | ((6 >> 8) << 5), 6 & 255])            # kind 01: len 4, offset 6
    tail = raw2[len(head) + 4:]
    lit2 = bytes([(len(tail) - 1) << 2]) + tail
    comp = ck._put_varint(len(raw2)) + lit + copy + lit2
    assert ck._snappy_decompress(comp) == raw2
    img = _hand_table([(b1, 0, b"applyx"), (comp, 1, b"b")])
    assert ck.read_table(img) == [(b"apple", b"1"), (b"apply", b"22"), (b"applyx", b""), (b"b", b"bananabana")]
    # a flipped byte in a block fails its checksum; a wrong magic is not a table
    bad = bytearray(img)
    bad[2] ^= 1
    with pytest.raises(ck.CheckpointError, match="checksum"):
        ck.read_table(bytes(bad))
    with pytest.raises(ck.CheckpointError, match="magic"):
        ck.read_table(img[:-1] + b"\x00")


def test_table_writer_roundtrip_many_blocks():
    rng = np.random.default_rng(0)
    keys = sorted({("layer_with_weights-%d/kernel/.ATTRIBUTES/%d" % (i % 37, i)).encode() for i in range(3000)})
    items = [(k, bytes(rng.integers(0, 256, size=int(rng.integers(0, 60))).astype(np.uint8))) for k in keys]
    img = ck.write_table(items, block_size=512)
    assert ck.read_table(img) == items
    assert img[-8:] == bytes.fromhex("57fb808b247547db") and len(ck.read_table(ck.write_table([]))) == 0
    with pytest.raises(ValueError):
        ck.write_table([(b"b", b""), (b"a", b"")])


def test_bundle_roundtrip_all_dtypes_and_corruption(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {
        "conv2d/kernel": rng.standard_normal((3, 3, 8, 16)).astype(np.float32),
        "a/scalar": np.asarray(7, dtype=np.int64),
        "a/empty": np.zeros((0, 4), np.float32),
        "a/half": rng.standard_normal((5,)).astype(np.float16),
        "a/bool": np.asarray([True, False, True]),
        "a/double": rng.standard_normal((2, 2)),
        "z/int32": np.arange(-3, 3, dtype=np.int32),
    }
    prefix = str(tmp_path / "ckpt" / "weights_step_10")
    w = ck.TensorBundleWriter(prefix)
    for k, v in tensors.items():
        w.add(k, v)
    w.add_strings("names", [b"alpha", b"", b"gamma" * 50], shape=(3,))
    w.finish()
    assert sorted(os.listdir(tmp_path / "ckpt")) == ["weights_step_10.data-00000-of-00001", "weights_step_10.index"]
    r = ck.TensorBundleReader(prefix)
    assert r.keys() == sorted(list(tensors) + ["names"]) and r.num_shards == 1
    for k, v in tensors.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape
        np.testing.assert_array_equal(got, v)
    assert r.get_tensor("names").tolist() == [b"alpha", b"", b"gamma" * 50]
    # the header entry is BundleHeaderProto{num_shards: 1, version{producer: 1}} = 08 01 1a 02 08 01
    with open(prefix + ".index", "rb") as f:
        table = ck.read_table(f.read())
    assert table[0] == (b"", bytes.fromhex("08011a020801"))
    # offsets follow the order of the add() calls, entries carry the masked CRC-32C of their bytes
    e = r.entries["conv2d/kernel"]
    assert e["offset"] == 0 and e["size"] == 3 * 3 * 8 * 16 * 4 and e["dtype"] == 1
    assert e["crc"] == _mask(ck._crc(tensors["conv2d/kernel"].tobytes()))
    # flip one data byte -> that tensor (only) fails its checksum
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    raw[10] ^= 0x01
    open(data, "wb").write(bytes(raw))
    r2 = ck.TensorBundleReader(prefix)
    with pytest.raises(ck.CheckpointError, match="checksum"):
        r2.get_tensor("conv2d/kernel")
    np.testing.assert_array_equal(r2.get_tensor("z/int32"), tensors["z/int32"])
    with pytest.raises(FileNotFoundError):
        ck.TensorBundleReader(str(tmp_path / "nope"))


def test_bfloat16_and_entry_protos_written_by_the_protobuf_runtime(tmp_path):
    """A bundle whose entry protos come from the official protobuf runtime (schema from tensor_bundle.proto /
    tensor_shape.proto), with a DT_BFLOAT16 tensor and a non-zero offset."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    f = descriptor_pb2.FileDescriptorProto(name="rnet_bundle.proto", package="rb", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto
    dim = descriptor_pb2.DescriptorProto(name="Dim")
    dim.field.add(name="size", number=1, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    dim.field.add(name="name", number=2, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    shape = f.message_type.add(name="TensorShapeProto")
    shape.nested_type.append(dim)
    shape.field.add(name="dim", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_REPEATED, type_name=".rb.TensorShapeProto.Dim")
    ent = f.message_type.add(name="BundleEntryProto")
    ent.field.add(name="dtype", number=1, type=T.TYPE_INT32, label=T.LABEL_OPTIONAL)
    ent.field.add(name="shape", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".rb.TensorShapeProto")
    ent.field.add(name="shard_id", number=3, type=T.TYPE_INT32, label=T.LABEL_OPTIONAL)
    ent.field.add(name="offset", number=4, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    ent.field.add(name="size", number=5, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    ent.field.add(name="crc32c", number=6, type=T.TYPE_FIXED32, label=T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    Entry = message_factory.GetMessageClass(pool.FindMessageTypeByName("rb.BundleEntryProto"))

    vals = np.float32([1.0, -2.5, 3.140625, 0.0, 65280.0, 1e-3])
    bf = (vals.view(np.uint32) >> 16).astype("<u2")
    pad = b"\xee" * 24
    payload = bf.tobytes()
    e = Entry(dtype=14, shard_id=0, offset=len(pad), size=len(payload), crc32c=_mask(ck._crc(payload)))
    e.shape.dim.add(size=2)
    e.shape.dim.add(size=3)
    prefix = str(tmp_path / "bf")
    open(prefix + ".data-00000-of-00001", "wb").write(pad + payload)
    open(prefix + ".index", "wb").write(ck.write_table([(b"", bytes.fromhex("08011a020801")), (b"w", e.SerializeToString())]))
    got = ck.TensorBundleReader(prefix).get_tensor("w")
    assert got.dtype == np.float32 and got.shape == (2, 3)
    np.testing.assert_array_equal(got.reshape(-1), (bf.astype(np.uint32) << 16).view(np.float32))
    # and the writer's own entry protos parse with the official runtime
    w = ck.TensorBundleWriter(str(tmp_path / "own"))
    w.add("first", np.zeros((4,), np.float32))
    w.add("second", np.ones((2, 0, 5), np.int64))
    w.add("third", np.arange(6, dtype=np.float32).reshape(2, 3))
    w.finish()
    table = dict(ck.read_table(open(str(tmp_path / "own") + ".index", "rb").read()))
    third = Entry()
    third.ParseFromString(table[b"third"])
    assert (third.dtype, third.offset, third.size, [d.size for d in third.shape.dim]) == (1, 16, 24, [2, 3])
    second = Entry()
    second.ParseFromString(table[b"second"])
    assert (second.dtype, second.offset, second.size, [d.size for d in second.shape.dim]) == (9, 16, 0, [2, 0, 5])


def _keras_style_object_graph():
    """root(0) -> layer_with_weights-0 (1) -> kernel (2), bias (3); root -> optimizer (4) -> iter (5); slot
    variables momentum (6) of kernel and average (7) of kernel hang off the optimizer node."""
    def ref(node_id, name):
        return ck._pb_bytes(1, ck._pb_varint(1, node_id) + ck._pb_bytes(2, name.encode()))

    def var(full, key):
        return ck._pb_bytes(2, ck._pb_bytes(1, b"VARIABLE_VALUE") + ck._pb_bytes(2, full.encode()) + ck._pb_bytes(3, key.encode()))

    def slot(orig, name, node):
        return ck._pb_bytes(3, ck._pb_varint(1, orig) + ck._pb_bytes(2, name.encode()) + ck._pb_varint(3, node))

    sfx = "/.ATTRIBUTES/VARIABLE_VALUE"
    nodes = [
        ref(1, "layer_with_weights-0") + ref(4, "optimizer"),
        ref(2, "kernel") + ref(3, "bias"),
        var("conv2d/kernel", "layer_with_weights-0/kernel" + sfx),
        var("conv2d/bias", "layer_with_weights-0/bias" + sfx),
        ref(5, "iter") + slot(2, "momentum", 6) + slot(2, "average", 7),
        var("SGD/iter", "optimizer/iter" + sfx),
        var("SGD/conv2d/kernel/momentum", "layer_with_weights-0/kernel/.OPTIMIZER_SLOT/optimizer/momentum" + sfx),
        var("conv2d/kernel/average", "layer_with_weights-0/kernel/.OPTIMIZER_SLOT/optimizer/average" + sfx),
    ]
    return b"".join(ck._pb_bytes(1, n) for n in nodes)


def test_keras_object_graph_maps_full_names_and_slots(tmp_path):
    rng = np.random.default_rng(2)
    kern, bias = rng.standard_normal((1, 1, 4, 8)).astype(np.float32), rng.standard_normal((8,)).astype(np.float32)
    mom, avg = kern * 0.1, kern * 0.9
    sfx = "/.ATTRIBUTES/VARIABLE_VALUE"
    prefix = str(tmp_path / "keras")
    w = ck.TensorBundleWriter(prefix)
    w.add_strings(ck.OBJECT_GRAPH_KEY, [_keras_style_object_graph()])
    w.add("layer_with_weights-0/bias" + sfx, bias)
    w.add("layer_with_weights-0/kernel" + sfx, kern)
    w.add("layer_with_weights-0/kernel/.OPTIMIZER_SLOT/optimizer/momentum" + sfx, mom)
    w.add("layer_with_weights-0/kernel/.OPTIMIZER_SLOT/optimizer/average" + sfx, avg)
    w.add("optimizer/iter" + sfx, np.asarray(1234, np.int64))
    w.finish()
    variables, slots = ck.load_weights(prefix)
    assert sorted(variables) == ["SGD/iter", "conv2d/bias", "conv2d/kernel"]     # slot variables are not model variables
    np.testing.assert_array_equal(variables["conv2d/kernel"], kern)
    np.testing.assert_array_equal(variables["conv2d/bias"], bias)
    assert int(variables["SGD/iter"]) == 1234
    assert sorted(slots) == [("conv2d/kernel", "average"), ("conv2d/kernel", "momentum")]
    np.testing.assert_array_equal(slots[("conv2d/kernel", "momentum")], mom)
    np.testing.assert_array_equal(slots[("conv2d/kernel", "average")], avg)


def test_save_load_weights_and_latest_checkpoint(tmp_path):
    rng = np.random.default_rng(3)
    variables = {f"conv2d_{i}/kernel": rng.standard_normal((3, 3, 4, 4)).astype(np.float32) for i in range(40)}
    variables["batch_normalization/moving_variance"] = rng.random((4,)).astype(np.float32)
    variables["SGD/iter"] = np.asarray(77, np.int64)
    slots = {(k, s): v * 0.5 for k, v in list(variables.items())[:5] for s in ("momentum", "average")}
    assert ck.latest_checkpoint(tmp_path) is None
    ck.save_weights(str(tmp_path / "weights_step_5"), variables, slots)
    ck.save_weights(str(tmp_path / "final_weights_step_9"), variables)
    assert ck.latest_checkpoint(tmp_path) == str(tmp_path / "final_weights_step_9")
    got, got_slots = ck.load_weights(str(tmp_path / "weights_step_5"))
    assert sorted(got) == sorted(variables)
    for k, v in variables.items():
        np.testing.assert_array_equal(got[k], v)
    assert sorted(got_slots) == sorted(slots)
    for k, v in slots.items():
        np.testing.assert_array_equal(got_slots[k], v)
    r = ck.TensorBundleReader(str(tmp_path / "weights_step_5"))
    assert "conv2d_3/kernel/.ATTRIBUTES/VARIABLE_VALUE" in r.keys()
    assert "conv2d_3/kernel/.OPTIMIZER_SLOT/optimizer/momentum/.ATTRIBUTES/VARIABLE_VALUE" in r.keys()
    # a TF1-style bundle (no object graph): keys are the variable names
    w = ck.TensorBundleWriter(str(tmp_path / "tf1"))
    w.add("resnet/conv1/weights", variables["conv2d_0/kernel"])
    w.finish()
    got, _ = ck.load_weights(str(tmp_path / "tf1"))
    assert list(got) == ["resnet/conv1/weights"]
