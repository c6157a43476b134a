"""Pure-Python reader/writer of TensorFlow's tensor-bundle (checkpoint v2) files — the on-disk layout
that ``tf.train.Saver.save/restore`` uses in the reference (model/base_model.py:223-243):

    <prefix>.index                 LevelDB-format table: key "" -> BundleHeaderProto, key <tensor
                                   name> -> BundleEntryProto{dtype, shape, shard_id, offset, size, crc32c}
    <prefix>.data-00000-of-00001   raw little-endian tensor bytes

Restated from the published formats (LevelDB table_format.md; tensorflow/core/protobuf/
tensor_bundle.proto; tensorflow/core/util/tensor_bundle/), no TensorFlow or protobuf dependency:
varints and the few proto fields are decoded by hand.  Only uncompressed blocks are supported (what
TF's BundleWriter emits); a snappy block raises.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DTYPE_CODES = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
                np.dtype("int64"): DT_INT64}

# ---- crc32c (Castagnoli), masked as LevelDB/TF do ------------------------------------------------
_CRC_TABLE: Optional[List[int]] = None


def _crc_table() -> List[int]:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    return _CRC_TABLE


def crc32c(data: bytes, crc: int = 0) -> int:
    tbl = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - 0xa282ead8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- varints / minimal protobuf --------------------------------------------------------------------
def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """field number -> list of raw values (int for varint/fixed, bytes for length-delimited)."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _get_varint(buf, pos)
        elif wire == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _get_varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type {}".format(wire))
        out.setdefault(field, []).append(val)
    return out


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _field(tag: int, wire: int) -> bytes:
    return _put_varint((tag << 3) | wire)


def _encode_entry(dtype_code: int, shape, offset: int, size: int, crc_masked: int) -> bytes:
    shp = b"".join(_field(2, 2) + _put_varint(len(d)) + d
                   for d in (_field(1, 0) + _put_varint(int(s)) for s in shape))
    out = _field(1, 0) + _put_varint(dtype_code)
    out += _field(2, 2) + _put_varint(len(shp)) + shp
    # shard_id = 0 is the proto default and omitted, like TF does
    if offset:
        out += _field(4, 0) + _put_varint(offset)
    out += _field(5, 0) + _put_varint(size)
    out += _field(6, 5) + struct.pack("<I", crc_masked)
    return out


def _encode_header() -> bytes:
    version = _field(1, 0) + _put_varint(1)                 # VersionDef.producer = 1
    return _field(1, 0) + _put_varint(1) + _field(3, 2) + _put_varint(len(version)) + version


# ---- LevelDB table ------------------------------------------------------------------------------------
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    contents = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(buf[offset:offset + size + 1]):
            raise ValueError("index block checksum mismatch")
    if ctype != 0:
        raise ValueError("compressed index block (type {}) not supported".format(ctype))
    return contents


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    out = []
    pos = 0
    key = b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(prefix: str, verify: bool = True) -> Dict[str, dict]:
    with open(prefix + ".index", "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError("{}.index is not a tensor-bundle index (bad magic)".format(prefix))
    footer = buf[-48:]
    pos = 0
    _, pos = _get_varint(footer, pos)        # metaindex handle
    _, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    entries: Dict[str, dict] = {}
    for _, handle in _block_entries(_read_block(buf, idx_off, idx_size, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        for key, val in _block_entries(_read_block(buf, boff, bsize, verify)):
            msg = _parse_proto(val)
            if key == b"":
                if msg.get(1, [1])[0] != 1:
                    raise ValueError("multi-shard bundles are not supported")
                if msg.get(2, [0])[0] != 0:
                    raise ValueError("big-endian bundles are not supported")
                continue
            shape = []
            if 2 in msg:
                for dim in _parse_proto(msg[2][0]).get(2, []):
                    shape.append(_signed64(_parse_proto(dim).get(1, [0])[0]))
            entries[key.decode()] = dict(dtype=msg.get(1, [0])[0], shape=tuple(shape),
                                         shard=msg.get(3, [0])[0], offset=msg.get(4, [0])[0],
                                         size=msg.get(5, [0])[0], crc=msg.get(6, [None])[0],
                                         sliced=7 in msg)
    return entries


def read_bundle(prefix: str, verify_data: bool = False) -> Dict[str, np.ndarray]:
    """name -> array for every float/int tensor in the bundle (partitioned variables are skipped)."""
    entries = read_index(prefix)
    out: Dict[str, np.ndarray] = {}
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        data = f.read()
    for name, e in entries.items():
        if e["sliced"] or e["dtype"] not in _DTYPES or e["shard"] != 0:
            continue
        raw = data[e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError("tensor {} runs past the end of the data file".format(name))
        if verify_data and e["crc"] is not None and unmask_crc(e["crc"]) != crc32c(raw):
            raise ValueError("tensor {} checksum mismatch".format(name))
        dt = _DTYPES[e["dtype"]]
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if n * dt.itemsize != e["size"]:
            raise ValueError("tensor {}: size {} does not match shape {}".format(name, e["size"], e["shape"]))
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    return out


def _build_block(entries: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out = bytearray()
    restarts = []
    last = b""
    for i, (key, val) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(last), len(key))
            while shared < m and last[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val))
        out += key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096) -> None:
    """Writes <prefix>.index + <prefix>.data-00000-of-00001 that TF's BundleReader (and read_bundle)
    accept: one shard, little-endian, uncompressed blocks, per-tensor masked crc32c."""
    names = sorted(tensors)                    # LevelDB tables need keys in bytewise order
    items: List[Tuple[bytes, bytes]] = [(b"", _encode_header())]
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as df:
        for name in names:
            arr = np.asarray(tensors[name])
            if arr.ndim and not arr.flags.c_contiguous:
                arr = np.ascontiguousarray(arr)
            code = _DTYPE_CODES[arr.dtype]
            raw = arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            df.write(raw)
            items.append((name.encode(), _encode_entry(code, arr.shape, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    out = bytearray()

    def emit(block: bytes) -> bytes:
        off = len(out)
        out.extend(block)
        out.append(0)                                                    # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    index_entries: List[Tuple[bytes, bytes]] = []
    cur: List[Tuple[bytes, bytes]] = []
    cur_bytes = 0
    for kv in items:
        cur.append(kv)
        cur_bytes += len(kv[0]) + len(kv[1]) + 8
        if cur_bytes >= block_size:
            index_entries.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index_entries.append((cur[-1][0], emit(_build_block(cur))))
    meta_handle = emit(_build_block([]))
    index_handle = emit(_build_block(index_entries, restart_interval=1))
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
