"""TensorFlow-free reader for TF "tensor bundle" checkpoints (model.ckpt.index + .data-00000-of-00001).

The reference restores weights with ``tf.train.Saver().restore(sess, modelPath/'model.ckpt')``
(reference UnMicst1-5.py:677-681, UnMicst2.py:656-660, UnMicst.py:510-514).  TensorFlow is a
third-party dependency that is absent here, so this module parses the published on-disk format
directly:

* ``model.ckpt.index`` is a LevelDB-format sorted string table (uncompressed blocks in every
  checkpoint the reference ships): 48-byte footer = metaindex handle + index handle (varint
  offset, varint size each), zero padding, little-endian magic 0xdb4775248b80fb57.  A block is a
  run of prefix-compressed entries ``(shared, non_shared, value_len | key_delta | value)`` followed
  by a restart array; every block is followed by a 1-byte compression type and a 4-byte CRC.
* data-block keys are variable names, values are ``BundleEntryProto`` messages
  {1: dtype, 2: TensorShapeProto{2: Dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c};
  the empty key holds the bundle header.
* ``model.ckpt.data-00000-of-00001`` holds the raw little-endian tensors at ``offset``.

Only what the UnMicst checkpoints need is implemented (float32 / int32 / int64 tensors, one shard).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np

_MAGIC = 0xDB4775248B80FB57

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 9: np.dtype("<i8")}


class CheckpointError(ValueError):
    """Raised for a malformed / unsupported checkpoint (the reference raises tf NotFoundError etc.)."""


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise CheckpointError("varint too long")


def _read_block(buf: bytes, offset: int, size: int):
    """Yield (key, value) pairs of one table block."""
    if offset + size + 5 > len(buf):
        raise CheckpointError("block handle outside file")
    if buf[offset + size] != 0:
        raise CheckpointError("compressed index blocks are not supported (type %d)" % buf[offset + size])
    block = buf[offset:offset + size]
    (num_restarts,) = struct.unpack_from("<I", block, size - 4)
    limit = size - 4 - 4 * num_restarts
    pos = 0
    key = b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        value_len, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        value = block[pos:pos + value_len]
        pos += value_len
        yield key, value


def _parse_proto(buf: bytes) -> Dict[int, list]:
    """Minimal protobuf wire decoder: {field: [raw values]} (varint -> int, len-delimited -> bytes)."""
    out: Dict[int, list] = {}
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wire)
        out.setdefault(field, []).append(val)
    return out


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for dim in _parse_proto(buf).get(2, []):
        d = _parse_proto(dim)
        dims.append(int(d.get(1, [0])[0]))
    return tuple(dims)


def read_index(index_path: str) -> Dict[str, dict]:
    """Return {variable name: {dtype, shape, shard, offset, size}} for every tensor in the bundle."""
    with open(index_path, "rb") as f:
        buf = f.read()
    if len(buf) < 48:
        raise CheckpointError("index file too short")
    footer = buf[-48:]
    (magic,) = struct.unpack_from("<Q", footer, 40)
    if magic != _MAGIC:
        raise CheckpointError("bad table magic %x" % magic)
    pos = 0
    _, pos = _varint(footer, pos)  # metaindex offset
    _, pos = _varint(footer, pos)  # metaindex size
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries: Dict[str, dict] = {}
    for _, handle in _read_block(buf, idx_off, idx_size):
        hpos = 0
        b_off, hpos = _varint(handle, hpos)
        b_size, hpos = _varint(handle, hpos)
        for key, value in _read_block(buf, b_off, b_size):
            if key == b"":
                continue  # BundleHeaderProto
            msg = _parse_proto(value)
            dtype = int(msg.get(1, [0])[0])
            shape = _parse_shape(msg[2][0]) if 2 in msg else ()
            entries[key.decode("utf-8")] = {
                "dtype": dtype,
                "shape": shape,
                "shard": int(msg.get(3, [0])[0]),
                "offset": int(msg.get(4, [0])[0]),
                "size": int(msg.get(5, [0])[0]),
            }
    return entries


def load_checkpoint(prefix: str) -> Dict[str, np.ndarray]:
    """Load every tensor of ``<prefix>.index`` / ``<prefix>.data-00000-of-00001`` as numpy arrays."""
    index_path = prefix + ".index"
    data_path = prefix + ".data-00000-of-00001"
    if not os.path.exists(index_path):
        raise FileNotFoundError(index_path)
    if not os.path.exists(data_path):
        raise FileNotFoundError(
            "%s is missing (the reference downloads the solo/duo weight shards at image build time, "
            "reference Dockerfile:5-6)" % data_path)
    entries = read_index(index_path)
    data = np.memmap(data_path, dtype=np.uint8, mode="r")
    out: Dict[str, np.ndarray] = {}
    for name, e in entries.items():
        if e["shard"] != 0:
            raise CheckpointError("multi-shard bundles are not supported (%s)" % name)
        if e["dtype"] not in _DTYPES:
            raise CheckpointError("unsupported dtype %d for %s" % (e["dtype"], name))
        dt = _DTYPES[e["dtype"]]
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if n * dt.itemsize != e["size"] or e["offset"] + e["size"] > data.size:
            raise CheckpointError("size mismatch for %s" % name)
        arr = np.frombuffer(data[e["offset"]:e["offset"] + e["size"]].tobytes(), dtype=dt)
        out[name] = arr.reshape(e["shape"]).copy()
    return out
