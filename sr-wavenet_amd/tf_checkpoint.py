"""TensorFlow checkpoint bundles (``tf.train.Saver``, format V2) without TensorFlow -- SURVEY section 8(f) rank 4.

The reference saves and restores with ``tf.train.Saver(self.network_params)`` (model.py:119, 217-235): a directory
holding ``checkpoint`` (text: ``model_checkpoint_path: "model.ckpt-N"``), ``model.ckpt-N.index`` and
``model.ckpt-N.data-00000-of-00001``.  This module reads such a bundle into ``{variable name: ndarray}`` and writes
one from such a dict, so weights trained with the reference can be loaded into these models by name (the engines'
``tf_variables()`` use the reference's variable names) and vice versa.

Format (TensorFlow ``tensor_bundle`` over its LevelDB-style ``table``; restated from the published format):

* ``.data-SSSSS-of-NNNNN``: the tensors' raw little-endian bytes back to back.
* ``.index``: an immutable sorted string table.  A *block* is a run of entries
  ``varint32 shared | varint32 non_shared | varint32 value_len | key[shared:] | value`` followed by an array of
  ``uint32`` restart offsets and their ``uint32`` count; on disk each block is trailed by one compression byte
  (0 = none, 1 = snappy) and the masked CRC-32C of block + that byte.  Data blocks hold the entries; the index block
  maps a separator key >= the last key of each data block to its ``BlockHandle`` (varint64 offset, varint64 size);
  the 48-byte footer holds the metaindex and index handles, zero padding and the magic ``0xdb4775248b80fb57``.
* key ``""`` -> ``BundleHeaderProto`` (num_shards = 1, endianness = 2, version = 3); key = tensor name ->
  ``BundleEntryProto`` (dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32, masked CRC-32C
  of the tensor's bytes), slices = 7).

**Parity unpinned**: the reference ships no checkpoint and TensorFlow is not installable here, so the reader is checked
against this module's own writer, hand-assembled blocks (prefix-compressed keys, several data blocks, restart arrays)
and corrupted inputs only.  It accepts what the format allows (shared key prefixes, any number of blocks and shards)
rather than only what the writer below emits; snappy-compressed index blocks (TensorFlow writes the bundle index
uncompressed) are rejected with a clear error.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from . import nsynth as _io

MAGIC = 0xDB4775248B80FB57
FOOTER_LEN = 48
BLOCK_TRAILER = 5

# DataType enum (tensorflow/core/framework/types.proto) <-> numpy
_DT_TO_NP = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
             6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
             22: np.dtype("<u4"), 23: np.dtype("<u8")}
_NP_TO_DT = {v: k for k, v in _DT_TO_NP.items()}
DT_BFLOAT16 = 14


class BundleError(RuntimeError):
    pass


def _crc_masked(buf: bytes) -> int:
    lib = _io.load()
    if not hasattr(lib, "_crc_bound"):
        lib.srwn_crc32c.restype, lib.srwn_crc32c.argtypes = C.c_uint32, [C.c_char_p, C.c_uint64]
        lib.srwn_crc32c_mask.restype, lib.srwn_crc32c_mask.argtypes = C.c_uint32, [C.c_uint32]
        lib._crc_bound = True
    return int(lib.srwn_crc32c_mask(lib.srwn_crc32c(buf, len(buf))))


# ---------------------------------------------------------------------------------------------- varints / protobuf
def _get_varint(b: bytes, p: int) -> Tuple[int, int]:
    v, s = 0, 0
    while True:
        if p >= len(b):
            raise BundleError("truncated varint")
        c = b[p]
        p += 1
        v |= (c & 0x7F) << s
        if not c & 0x80:
            return v, p
        s += 7
        if s > 63:
            raise BundleError("varint longer than 64 bits")


def _put_varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        c = v & 0x7F
        v >>= 7
        out.append(c | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_fields(b: bytes) -> Iterable[Tuple[int, int, object]]:
    """(field number, wire type, value) of one protobuf message; value = int (varint, fixed) or bytes."""
    p = 0
    while p < len(b):
        key, p = _get_varint(b, p)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _get_varint(b, p)
        elif wt == 1:
            if p + 8 > len(b):
                raise BundleError("truncated fixed64")
            v, p = struct.unpack_from("<Q", b, p)[0], p + 8
        elif wt == 2:
            n, p = _get_varint(b, p)
            if p + n > len(b):
                raise BundleError("truncated length-delimited field")
            v, p = b[p:p + n], p + n
        elif wt == 5:
            if p + 4 > len(b):
                raise BundleError("truncated fixed32")
            v, p = struct.unpack_from("<I", b, p)[0], p + 4
        else:
            raise BundleError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(b: bytes) -> List[int]:
    dims = []
    for f, wt, v in _pb_fields(b):
        if f == 2 and wt == 2:                      # repeated Dim
            size = 0
            for g, gw, gv in _pb_fields(v):
                if g == 1 and gw == 0:
                    size = _signed64(gv)
            dims.append(size)
        elif f == 3 and wt == 0 and v:
            raise BundleError("tensor of unknown rank in checkpoint")
    return dims


def _parse_entry(b: bytes) -> dict:
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, slices=False)
    for f, wt, v in _pb_fields(b):
        if f == 1 and wt == 0:
            e["dtype"] = v
        elif f == 2 and wt == 2:
            e["shape"] = _parse_shape(v)
        elif f == 3 and wt == 0:
            e["shard_id"] = v
        elif f == 4 and wt == 0:
            e["offset"] = _signed64(v)
        elif f == 5 and wt == 0:
            e["size"] = _signed64(v)
        elif f == 6 and wt == 5:
            e["crc32c"] = v
        elif f == 7:
            e["slices"] = True
    return e


def _parse_header(b: bytes) -> dict:
    h = dict(num_shards=1, endianness=0)
    for f, wt, v in _pb_fields(b):
        if f == 1 and wt == 0:
            h["num_shards"] = v
        elif f == 2 and wt == 0:
            h["endianness"] = v
    return h


# ---------------------------------------------------------------------------------------------- table reader
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset < 0 or size < 0 or offset + size + BLOCK_TRAILER > len(data):
        raise BundleError("block handle (%d, %d) outside the index file" % (offset, size))
    body = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        want = struct.unpack_from("<I", data, offset + size + 1)[0]
        if _crc_masked(data[offset:offset + size + 1]) != want:
            raise BundleError("index block at %d: CRC mismatch" % offset)
    if ctype == 1:
        raise BundleError("snappy-compressed index block (not supported; TensorFlow writes the bundle index uncompressed)")
    if ctype != 0:
        raise BundleError("unknown block compression type %d" % ctype)
    return body


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise BundleError("block shorter than its restart count")
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    if end < 0:
        raise BundleError("block restart array larger than the block")
    out, p, key = [], 0, b""
    while p < end:
        shared, p = _get_varint(block, p)
        non_shared, p = _get_varint(block, p)
        vlen, p = _get_varint(block, p)
        if shared > len(key) or p + non_shared + vlen > end:
            raise BundleError("corrupt block entry")
        key = key[:shared] + block[p:p + non_shared]
        p += non_shared
        out.append((key, block[p:p + vlen]))
        p += vlen
    return out


def read_index(path: str, verify: bool = True) -> Tuple[dict, Dict[str, dict]]:
    """``prefix.index`` -> (header, {tensor name: entry dict})."""
    data = open(path, "rb").read()
    if len(data) < FOOTER_LEN:
        raise BundleError("%s: shorter than a table footer" % path)
    footer = data[-FOOTER_LEN:]
    if struct.unpack_from("<Q", footer, FOOTER_LEN - 8)[0] != MAGIC:
        raise BundleError("%s: not a TensorFlow checkpoint index (bad table magic)" % path)
    p = 0
    _, p = _get_varint(footer, p)          # metaindex handle (unused)
    _, p = _get_varint(footer, p)
    ioff, p = _get_varint(footer, p)
    isize, p = _get_varint(footer, p)
    header, entries = None, {}
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, q = _get_varint(handle, 0)
        bsize, q = _get_varint(handle, q)
        for key, value in _block_entries(_read_block(data, boff, bsize, verify)):
            if key == b"":
                header = _parse_header(value)
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    if header is None:
        raise BundleError("%s: no bundle header entry" % path)
    if header["endianness"] != 0:
        raise BundleError("%s: big-endian bundle" % path)
    return header, entries


def read_bundle(prefix: str, names: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    """All (or the named) tensors of the checkpoint ``prefix`` (``prefix.index`` + ``prefix.data-*``).  bfloat16
    tensors come back as float32."""
    header, entries = read_index(prefix + ".index", verify)
    want = set(entries) if names is None else set(names)
    missing = want - set(entries)
    if missing:
        raise KeyError("not in checkpoint %s: %s" % (prefix, sorted(missing)[:5]))
    shards: Dict[int, np.memmap] = {}
    out = {}
    for name in sorted(want):
        e = entries[name]
        if e["slices"]:
            raise BundleError("%s: partitioned variable (tensor slices) not supported" % name)
        sid = e["shard_id"]
        if sid not in shards:
            sp = "%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"])
            if not os.path.exists(sp):
                raise BundleError("missing data shard %s" % sp)
            shards[sid] = np.memmap(sp, dtype=np.uint8, mode="r")
        raw = shards[sid]
        if e["offset"] < 0 or e["size"] < 0 or e["offset"] + e["size"] > raw.shape[0]:
            raise BundleError("%s: bytes [%d, +%d) outside its data shard" % (name, e["offset"], e["size"]))
        buf = bytes(raw[e["offset"]:e["offset"] + e["size"]])
        if verify and e["crc32c"] is not None and _crc_masked(buf) != e["crc32c"]:
            raise BundleError("%s: tensor CRC mismatch" % name)
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if e["dtype"] == DT_BFLOAT16:
            if len(buf) != 2 * count:
                raise BundleError("%s: %d bytes for %d bfloat16 values" % (name, len(buf), count))
            arr = (np.frombuffer(buf, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
        else:
            if e["dtype"] not in _DT_TO_NP:
                raise BundleError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
            dt = _DT_TO_NP[e["dtype"]]
            if len(buf) != dt.itemsize * count:
                raise BundleError("%s: %d bytes for shape %s of %s" % (name, len(buf), e["shape"], dt))
            arr = np.frombuffer(buf, dtype=dt)
        out[name] = arr.reshape(e["shape"]).copy()
    return out


# ---------------------------------------------------------------------------------------------- writer
def _pb_varint_field(f: int, v: int) -> bytes:
    return _put_varint(f << 3) + _put_varint(v)


def _pb_bytes_field(f: int, b: bytes) -> bytes:
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + b


def _entry_proto(dtype: int, shape, offset: int, size: int, crc: int) -> bytes:
    shp = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)
    out = _pb_varint_field(1, dtype) + _pb_bytes_field(2, shp)
    if offset:
        out += _pb_varint_field(4, offset)
    out += _pb_varint_field(5, size)
    out += _put_varint((6 << 3) | 5) + struct.pack("<I", crc)
    return out


def _build_block(items: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            n = min(len(prev), len(k))
            while shared < n and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return bytes(out)


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray], block_size: int = 4096) -> None:
    """Writes ``prefix.index`` and ``prefix.data-00000-of-00001`` holding ``tensors`` (one shard, uncompressed index,
    data blocks of about ``block_size`` bytes, keys prefix-compressed with 16-entry restart intervals)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items: List[Tuple[bytes, bytes]] = []
    header = _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1))   # num_shards 1, little endian, version producer 1
    items.append((b"", header))
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            a = np.asarray(tensors[name])
            a = a if a.flags.c_contiguous else a.copy()          # (ascontiguousarray would turn a scalar into [1])
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            if np.dtype(dt) not in _NP_TO_DT:
                raise BundleError("%s: dtype %s has no checkpoint encoding here" % (name, a.dtype))
            buf = a.astype(dt, copy=False).tobytes()
            f.write(buf)
            items.append((name.encode("utf-8"), _entry_proto(_NP_TO_DT[np.dtype(dt)], a.shape, offset, len(buf), _crc_masked(buf))))
            offset += len(buf)
    out = bytearray()

    def emit(block: bytes) -> bytes:
        handle = _put_varint(len(out)) + _put_varint(len(block))
        out.extend(block + b"\x00")
        out.extend(struct.pack("<I", _crc_masked(block + b"\x00")))
        return handle

    index_items, cur, cur_bytes = [], [], 0
    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            index_items.append((cur[-1][0], emit(_build_block(cur))))
            cur, cur_bytes = [], 0
    if cur:
        index_items.append((cur[-1][0], emit(_build_block(cur))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_items, restart_interval=1))
    footer = meta + index
    footer += b"\x00" * (FOOTER_LEN - 8 - len(footer)) + struct.pack("<Q", MAGIC)
    out.extend(footer)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


# ---------------------------------------------------------------------------------------------- Saver directory
def latest_checkpoint(logdir: str) -> Optional[str]:
    """The path prefix named by ``logdir/checkpoint`` (tf.train.get_checkpoint_state(...).model_checkpoint_path)."""
    state = os.path.join(logdir, "checkpoint")
    if not os.path.exists(state):
        return None
    for line in open(state):
        if line.startswith("model_checkpoint_path:"):
            path = line.split(":", 1)[1].strip().strip('"')
            return path if os.path.isabs(path) else os.path.join(logdir, path)
    return None


def write_checkpoint_state(logdir: str, name: str) -> None:
    with open(os.path.join(logdir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))


def is_bundle(prefix: str) -> bool:
    return os.path.exists(prefix + ".index")
