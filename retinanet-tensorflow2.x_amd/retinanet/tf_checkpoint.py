"""TensorFlow checkpoints without TensorFlow — SURVEY §8(f)-3.

The reference writes / reads its weights with Keras `model.save_weights` / `load_weights` in the TF2 object-graph
format (executor.py:119, 221-244, 652-654, 695-697; resnet.py:404-405, efficientnet.py:1044-1045) and finds the
newest one with `tf.train.latest_checkpoint` (the `checkpoint` state file).  This module reads and writes that
on-disk format directly:

  <prefix>.index                 an SSTable (LevelDB table format): key "" -> BundleHeaderProto, every other key ->
                                 BundleEntryProto {dtype, shape, shard_id, offset, size, masked crc32c}
  <prefix>.data-00000-of-00001   the tensors' raw little-endian bytes at those offsets
  key `_CHECKPOINTABLE_OBJECT_GRAPH`   a scalar string tensor holding the TrackableObjectGraph proto: nodes with
                                 named children, per-variable attributes {name, full_name, checkpoint_key} and slot
                                 variable references (optimizer `momentum`, EMA `average`)

PARITY UNPINNED: TensorFlow is not installable here and the reference ships no checkpoint file, so the formats are
restated from their public definitions (LevelDB table_format.md, tensor_bundle.proto, trackable_object_graph.proto,
tensor_bundle.cc's string-tensor layout); the tests pin reader against writer, the CRCs against RFC 3720, and the
table layer against hand-built blocks (prefix compression, restarts, snappy).  Block CRCs, tensor CRCs and sizes are
all verified on read, so a layout misunderstanding fails loudly instead of loading garbage.
"""
from __future__ import annotations

import os
import struct

import numpy as np

from retinanet import _C

OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
_VALUE_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"
_TABLE_MAGIC = 0xDB4775248B80FB57

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
DT_STRING, DT_BFLOAT16 = 7, 14
_DTYPE_IDS = {v: k for k, v in _DTYPES.items()}


class CheckpointError(IOError):
    pass


def _crc(data):
    a = np.frombuffer(data, dtype=np.uint8)
    return _C.lib().rn_crc32c(a.ctypes.data if a.size else None, a.size)


def _mask(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


class _Crc:
    """crc32c::Extend over several pieces (pieces are concatenated; checkpoints are not a hot path)."""

    def __init__(self):
        self.parts = []

    def extend(self, b):
        self.parts.append(bytes(b))

    def value(self):
        return _crc(b"".join(self.parts))


# ---- varints / a minimal protobuf wire codec ---------------------------------------------------------------
def _get_varint(buf, pos):
    v, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise CheckpointError("varint too long")


def _put_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _pb_decode(buf):
    """{field number: [values]}; varint -> int, fixed32/64 -> int, length-delimited -> bytes."""
    out, pos, buf = {}, 0, bytes(buf)
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _get_varint(buf, pos)
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _get_varint(buf, pos)
            if pos + n > len(buf):
                raise CheckpointError("truncated protobuf field")
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError(f"unsupported protobuf wire type {wire}")
        out.setdefault(field, []).append(v)
    return out


def _pb_varint(field, v):
    return _put_varint(field << 3) + _put_varint(v)


def _pb_bytes(field, b):
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + bytes(b)


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- snappy (index blocks may be compressed by other writers; TensorFlow's bundle writer does not) -------------
def _snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):          # byte-wise: copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


# ---- LevelDB table ---------------------------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    if offset + size + 5 > len(data):
        raise CheckpointError("block handle past the end of the index file")
    body, ctype = data[offset:offset + size], data[offset + size]
    if verify:
        want = struct.unpack_from("<I", data, offset + size + 1)[0]
        if _mask(_crc(data[offset:offset + size + 1])) != want:
            raise CheckpointError("index block checksum mismatch")
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise CheckpointError(f"unknown block compression {ctype}")
    return body


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError("block too small")
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 * (nrestart + 1)
    if end < 0:
        raise CheckpointError("bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + unshared + vlen > end:
            raise CheckpointError("corrupt block entry")
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(data, verify=True):
    """All (key, value) pairs of an SSTable image, in key order."""
    data = bytes(data)
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != _TABLE_MAGIC:
        raise CheckpointError("not an SSTable (bad magic)")
    footer = data[-48:]
    _, p = _get_varint(footer, 0)          # metaindex handle (unused)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        off, q = _get_varint(handle, 0)
        size, q = _get_varint(handle, q)
        out.extend(_block_entries(_read_block(data, off, size, verify)))
    return out


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(self.last), len(key))
            while shared < m and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * (len(self.restarts) + 1)

    def empty(self):
        return not self.buf


def write_table(items, block_size=4096):
    """SSTable image of sorted (key, value) byte pairs: uncompressed data blocks (prefix compression, restart
    interval 16), an empty metaindex block, an index block with one entry per data block, the 48-byte footer."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                               # kNoCompression
        out.extend(struct.pack("<I", _mask(_crc(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    index = _BlockBuilder(1)
    cur = _BlockBuilder(16)
    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError("table keys must be strictly increasing")
        cur.add(key, value)
        prev = key
        if cur.size() >= block_size:
            index.add(key, emit(cur.finish()))
            cur = _BlockBuilder(16)
    if not cur.empty():
        index.add(prev, emit(cur.finish()))
    meta = emit(_BlockBuilder(16).finish())
    idx = emit(index.finish())
    footer = meta + idx
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC))
    return bytes(out)


# ---- tensor bundle ---------------------------------------------------------------------------------------------
def _decode_shape(b):
    dims = []
    for d in _pb_decode(b).get(2, []):
        dims.append(_signed(_pb_decode(d).get(1, [0])[0]))
    return tuple(dims)


def _encode_shape(shape):
    return b"".join(_pb_bytes(2, _pb_varint(1, int(d)) if d else b"") for d in shape)


def _bf16_to_f32(raw):
    return (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)


class TensorBundleReader:
    """`tf.train.load_checkpoint(prefix)`-like access: keys(), shape/dtype, get_tensor()."""

    def __init__(self, prefix, verify=True):
        self.prefix = str(prefix)
        self.verify = verify
        if not os.path.exists(self.prefix + ".index"):
            raise FileNotFoundError(self.prefix + ".index")
        with open(self.prefix + ".index", "rb") as f:
            table = read_table(f.read(), verify)
        if not table or table[0][0] != b"":
            raise CheckpointError("bundle header entry missing")
        hdr = _pb_decode(table[0][1])
        self.num_shards = hdr.get(1, [0])[0]
        if hdr.get(2, [0])[0] != 0:
            raise CheckpointError("big-endian bundles are not supported")
        self.entries = {}
        for k, v in table[1:]:
            e = _pb_decode(v)
            if 7 in e:
                raise CheckpointError(f"{k.decode()}: sliced (partitioned) variables are not supported")
            self.entries[k.decode()] = dict(dtype=e.get(1, [0])[0], shape=_decode_shape(e[2][0]) if 2 in e else (),
                                            shard=e.get(3, [0])[0], offset=_signed(e.get(4, [0])[0]),
                                            size=_signed(e.get(5, [0])[0]), crc=e.get(6, [0])[0])
        self._shards = {}

    def keys(self):
        return sorted(self.entries)

    def has_tensor(self, key):
        return key in self.entries

    def _raw(self, e):
        sid = e["shard"]
        if sid not in self._shards:
            path = f"{self.prefix}.data-{sid:05d}-of-{self.num_shards:05d}"
            self._shards[sid] = np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, np.uint8)
        data = self._shards[sid]
        if e["offset"] + e["size"] > data.size:
            raise CheckpointError("tensor past the end of the data shard")
        return data[e["offset"]:e["offset"] + e["size"]]

    def get_tensor(self, key):
        e = self.entries[key]
        raw = self._raw(e)
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if e["dtype"] == DT_STRING:
            return self._strings(raw, n, e).reshape(e["shape"]) if e["shape"] else self._strings(raw, n, e)[0]
        if self.verify and _mask(_crc(raw)) != e["crc"]:
            raise CheckpointError(f"{key}: tensor checksum mismatch")
        if e["dtype"] == DT_BFLOAT16:
            arr = _bf16_to_f32(raw)
        elif e["dtype"] in _DTYPES:
            arr = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]])
        else:
            raise CheckpointError(f"{key}: unsupported dtype enum {e['dtype']}")
        if arr.size != n:
            raise CheckpointError(f"{key}: {arr.size} elements for shape {e['shape']}")
        return arr.reshape(e["shape"]).copy()

    def _strings(self, raw, n, e):
        raw = bytes(raw)
        lens, pos = [], 0
        crc = _Crc()
        for _ in range(n):
            v, pos = _get_varint(raw, pos)
            lens.append(v)
            crc.extend(struct.pack("<I", v) if v <= 0xFFFFFFFF else struct.pack("<Q", v))
        cks = raw[pos:pos + 4]
        if self.verify and struct.unpack("<I", cks)[0] != _mask(crc.value()):
            raise CheckpointError("string tensor length checksum mismatch")
        crc.extend(cks)
        pos += 4
        out = np.empty((n,), dtype=object)
        for i, ln in enumerate(lens):
            out[i] = raw[pos:pos + ln]
            crc.extend(out[i])
            pos += ln
        if pos != len(raw):
            raise CheckpointError("string tensor size mismatch")
        if self.verify and _mask(crc.value()) != e["crc"]:
            raise CheckpointError("string tensor checksum mismatch")
        return out

    # -- object graph --------------------------------------------------------------------------------------------
    def object_graph(self):
        """(variables, slots): variables = {full_name: checkpoint_key}; slots = {(full_name, slot_name): key}."""
        if OBJECT_GRAPH_KEY not in self.entries:
            return {}, {}
        nodes = [_pb_decode(n) for n in _pb_decode(self.get_tensor(OBJECT_GRAPH_KEY)).get(1, [])]
        attr = {}
        for i, nd in enumerate(nodes):
            for a in nd.get(2, []):
                a = _pb_decode(a)
                if a.get(1, [b""])[0] == b"VARIABLE_VALUE":
                    attr[i] = (a.get(2, [b""])[0].decode(), a.get(3, [b""])[0].decode())
        slots, slot_nodes = {}, set()
        for nd in nodes:
            for s in nd.get(3, []):
                s = _pb_decode(s)
                orig, slot = s.get(1, [0])[0], s.get(3, [0])[0]
                slot_nodes.add(slot)
                if orig in attr and slot in attr:
                    slots[(attr[orig][0], s.get(2, [b""])[0].decode())] = attr[slot][1]
        variables = {}
        for i, (full, key) in attr.items():
            if full and i not in slot_nodes:
                if full in variables and variables[full] != key:
                    # two different variables under one Keras name: loading by name cannot tell them apart
                    raise CheckpointError(f"object graph holds two variables named {full!r} "
                                          f"({variables[full]!r} and {key!r}): cannot be loaded by name")
                variables[full] = key
        return variables, slots


class TensorBundleWriter:
    """Writes `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard), tensors in the order added."""

    def __init__(self, prefix):
        self.prefix = str(prefix)
        self.entries = {}
        os.makedirs(os.path.dirname(os.path.abspath(self.prefix)), exist_ok=True)
        self._data = open(self.prefix + ".data-00000-of-00001.tmp", "wb")
        self._off = 0

    def _entry(self, key, dtype, shape, size, crc):
        if key in self.entries or key == "":
            raise ValueError(f"duplicate or empty key {key!r}")
        e = _pb_varint(1, dtype) + _pb_bytes(2, _encode_shape(shape))
        if self._off:
            e += _pb_varint(4, self._off)
        e += _pb_varint(5, size) + _put_varint((6 << 3) | 5) + struct.pack("<I", _mask(crc))
        self.entries[key] = e
        self._off += size

    def add(self, key, array):
        a = np.asarray(array)             # tobytes() is C order whatever the strides; 0-d stays 0-d
        dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
        if np.dtype(dt) not in _DTYPE_IDS:
            raise ValueError(f"{key}: unsupported dtype {a.dtype}")
        raw = a.astype(dt, copy=False).tobytes()
        self._data.write(raw)
        self._entry(key, _DTYPE_IDS[np.dtype(dt)], a.shape, len(raw), _crc(raw))

    def add_strings(self, key, values, shape=()):
        values = [bytes(v) for v in values]
        crc = _Crc()
        head = bytearray()
        for v in values:
            head += _put_varint(len(v))
            crc.extend(struct.pack("<I", len(v)) if len(v) <= 0xFFFFFFFF else struct.pack("<Q", len(v)))
        cks = struct.pack("<I", _mask(crc.value()))
        crc.extend(cks)
        for v in values:
            crc.extend(v)
        blob = bytes(head) + cks + b"".join(values)
        self._data.write(blob)
        self._entry(key, DT_STRING, shape, len(blob), crc.value())

    def finish(self):
        self._data.close()
        header = _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))      # num_shards 1, little endian, version.producer 1
        items = [(b"", header)] + [(k.encode(), v) for k, v in sorted(self.entries.items(), key=lambda kv: kv[0].encode())]
        with open(self.prefix + ".index.tmp", "wb") as f:
            f.write(write_table(items))
        os.replace(self.prefix + ".data-00000-of-00001.tmp", self.prefix + ".data-00000-of-00001")
        os.replace(self.prefix + ".index.tmp", self.prefix + ".index")


# ---- Keras-style weights on top of the bundle ----------------------------------------------------------------
def _object_graph_proto(var_keys, slot_keys):
    """root -> one child per variable (local name = variable name), slot variables as further nodes referenced from
    the root's slot_variables.  Readable by TensorFlow's checkpoint reader; Keras' own structural matching needs
    Keras' layer numbering, which cannot be derived without TensorFlow."""
    names = list(var_keys)
    node_of = {n: i + 1 for i, n in enumerate(names)}
    root = b"".join(_pb_bytes(1, _pb_varint(1, node_of[n]) + _pb_bytes(2, n.encode())) for n in names)
    nodes = []
    for n in names:
        nodes.append(_pb_bytes(2, _pb_bytes(1, b"VARIABLE_VALUE") + _pb_bytes(2, n.encode()) + _pb_bytes(3, var_keys[n].encode())))
    for (n, slot), key in slot_keys.items():
        nid = len(nodes) + 1
        nodes.append(_pb_bytes(2, _pb_bytes(1, b"VARIABLE_VALUE") + _pb_bytes(2, f"{n}/{slot}".encode()) + _pb_bytes(3, key.encode())))
        root += _pb_bytes(3, _pb_varint(1, node_of[n]) + _pb_bytes(2, slot.encode()) + _pb_varint(3, nid))
    return b"".join(_pb_bytes(1, nd) for nd in [root] + nodes)


def save_weights(prefix, variables, slots=None, update_state=True):
    """variables: {name: array} (conv kernels HWIO, as Keras holds them); slots: {(name, slot_name): array}.

    ONE-WAY INTEROP: the files are valid TensorFlow checkpoints (tf.train.load_checkpoint / list_variables read every
    tensor by key, and this module's reader maps them back through the object graph's full names), but the object
    graph written here is flat — root -> one child per variable name.  Keras' `model.load_weights` /
    `tf.train.Checkpoint.restore` match a checkpoint STRUCTURALLY (`layer_with_weights-N/...` children in Keras' own
    layer numbering), which cannot be reproduced without TensorFlow, so they will not restore these files into the
    reference's Keras model.  Reading checkpoints the reference wrote works the other way round (object_graph())."""
    slots = slots or {}
    var_keys = {n: n + _VALUE_SUFFIX for n in variables}
    slot_keys = {(n, s): f"{n}/.OPTIMIZER_SLOT/optimizer/{s}{_VALUE_SUFFIX}" for (n, s) in slots}
    w = TensorBundleWriter(prefix)
    w.add_strings(OBJECT_GRAPH_KEY, [_object_graph_proto(var_keys, slot_keys)])
    for n, a in variables.items():
        w.add(var_keys[n], np.asarray(a))
    for k, a in slots.items():
        w.add(slot_keys[k], np.asarray(a))
    w.finish()
    if update_state:
        _update_checkpoint_state(prefix)


def load_weights(prefix, verify=True):
    """-> (variables {name: array}, slots {(name, slot): array}) through the object graph's full names; a bundle
    without an object graph (TF1 name-based checkpoint) maps its keys directly."""
    r = TensorBundleReader(prefix, verify)
    var_keys, slot_keys = r.object_graph()
    if not var_keys:
        var_keys = {k: k for k in r.keys() if k != OBJECT_GRAPH_KEY}
    strip = lambda n: n[:-2] if n.endswith(":0") else n
    variables = {strip(n): r.get_tensor(k) for n, k in var_keys.items() if r.has_tensor(k)}
    slots = {(strip(n), s): r.get_tensor(k) for (n, s), k in slot_keys.items() if r.has_tensor(k)}
    return variables, slots


def _update_checkpoint_state(prefix):
    d, name = os.path.split(os.path.abspath(prefix))
    with open(os.path.join(d, "checkpoint.tmp"), "w") as f:
        f.write(f'model_checkpoint_path: "{name}"\nall_model_checkpoint_paths: "{name}"\n')
    os.replace(os.path.join(d, "checkpoint.tmp"), os.path.join(d, "checkpoint"))


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: the `model_checkpoint_path` of `<dir>/checkpoint` (relative paths resolve
    against the directory), or None."""
    state = os.path.join(str(checkpoint_dir), "checkpoint")
    if not os.path.exists(state):
        return None
    with open(state) as f:
        for line in f:
            line = line.strip()
            if line.startswith("model_checkpoint_path:"):
                p = line.split(":", 1)[1].strip().strip('"')
                p = p if os.path.isabs(p) else os.path.join(str(checkpoint_dir), p)
                return p if os.path.exists(p + ".index") else None
    return None
