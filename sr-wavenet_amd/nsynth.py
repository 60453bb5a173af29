"""Host-side mirror of the reference's ``nsynth.NsynthDataReader`` (nsynth.py:5-50) without TensorFlow.

Same constructor and ``next()``; records are parsed by the C++ reader behind ``include/srwn_io.h``
(``libsrwn_io.so``: mmap + record index + CRC-32C + a minimal protobuf wire reader, batches decoded by a thread pool).

Dataset semantics follow the reference's pipeline (nsynth.py:39-45): ``map -> shuffle(buffer 10000) -> repeat ->
batch``: a shuffle buffer of 10 000 records drawn uniformly (own RNG -- TF's shuffle order is not reproducible
either), repetition BEFORE batching (batches run across epoch boundaries), and without ``repeat`` a short final batch
followed by ``StopIteration`` (the reference raises ``tf.errors.OutOfRangeError``, filter_tfrecord.py:62).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsrwn_io.so")

# nsynth.py:10-25: name -> (kind, fixed length); kind 1 bytes (scalar string), 2 float, 3 int64
FEATURES = {
    "sample_rate": (3, 1), "note_str": (1, None), "qualities": (3, 10), "audio": (2, "audio_max_length"),
    "instrument_family": (3, 1), "pitch": (3, 1), "instrument_source": (3, 1), "instrument_str": (1, None),
    "instrument_source_str": (1, None), "note": (3, 1), "instrument": (3, 1), "instrument_family_str": (1, None),
    "velocity": (3, 1),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libsrwn_io.so not found at %s: build it with `python sr-wavenet_amd/build.py`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        lib.srwn_io_last_error.restype = C.c_char_p
        lib.srwn_tfr_open.restype, lib.srwn_tfr_open.argtypes = p, [C.c_char_p, i32]
        lib.srwn_tfr_close.restype, lib.srwn_tfr_close.argtypes = None, [p]
        lib.srwn_tfr_count.restype, lib.srwn_tfr_count.argtypes = i64, [p]
        lib.srwn_tfr_feature.restype, lib.srwn_tfr_feature.argtypes = C.c_int, [p, i64, C.c_char_p, p, p]
        for n in ("srwn_tfr_read_floats", "srwn_tfr_read_int64s", "srwn_tfr_read_bytes"):
            getattr(lib, n).restype = C.c_int
            getattr(lib, n).argtypes = [p, i64, C.c_char_p, p, i64, p]
        lib.srwn_tfr_read_batch.restype = C.c_int
        lib.srwn_tfr_read_batch.argtypes = [p, p, i32, C.c_char_p, i64, i32, p, C.c_char_p, p, i32]
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("TFRecord reader (code %d): %s" % (rc, load().srwn_io_last_error().decode()))


class TFRecordFile:
    """Random access to the ``tf.train.Example`` records of one TFRecord file."""

    def __init__(self, path: str, verify_crc: bool = True):
        self._h = load().srwn_tfr_open(os.fsencode(path), int(verify_crc))
        if not self._h:
            raise RuntimeError("cannot read %s: %s" % (path, load().srwn_io_last_error().decode()))
        self.path = path

    def __len__(self):
        return int(load().srwn_tfr_count(self._h))

    def close(self):
        if getattr(self, "_h", None):
            load().srwn_tfr_close(self._h)
            self._h = None

    __del__ = close

    def feature(self, idx: int, key: str):
        """The feature's values: float32 / int64 array, or bytes for a bytes_list."""
        kind, cnt = C.c_int32(), C.c_int64()
        _check(load().srwn_tfr_feature(self._h, idx, key.encode(), C.byref(kind), C.byref(cnt)))
        n = C.c_int64()
        if kind.value == 2:
            out = np.empty(cnt.value, np.float32)
            _check(load().srwn_tfr_read_floats(self._h, idx, key.encode(), out.ctypes.data, cnt.value, C.byref(n)))
            return out
        if kind.value == 3:
            out = np.empty(cnt.value, np.int64)
            _check(load().srwn_tfr_read_int64s(self._h, idx, key.encode(), out.ctypes.data, cnt.value, C.byref(n)))
            return out
        if kind.value == 1:
            _check(load().srwn_tfr_read_bytes(self._h, idx, key.encode(), None, 0, C.byref(n)))
            buf = C.create_string_buffer(max(int(n.value), 1))
            _check(load().srwn_tfr_read_bytes(self._h, idx, key.encode(), buf, n.value, C.byref(n)))
            return buf.raw[:n.value]
        return np.empty(0, np.float32)

    def batch(self, indices, num_samples: int, audio_len: int = 0, label_key: Optional[str] = "pitch", threads: int = 8):
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        audio = np.empty((len(idx), num_samples), np.float32)
        label = np.empty(len(idx), np.int64)
        _check(load().srwn_tfr_read_batch(self._h, idx.ctypes.data, len(idx), b"audio", int(audio_len), int(num_samples),
                                          audio.ctypes.data, label_key.encode() if label_key else None,
                                          label.ctypes.data if label_key else None, int(threads)))
        return audio, label


class NsynthDataReader(object):
    """nsynth.py:5-50.  ``next()`` -> ``(audio [B, num_samples] float32, pitch one-hot [B, 128] float32)`` in reduced
    mode, else a dict of the 13 parsed features (batched like ``tf.parse_single_example`` + ``batch``)."""

    def __init__(self, filepath, batch_size, num_samples=16000, reduced=True, shuffle=True, repeat=True,
                 audio_max_length=64000, seed=None, verify_crc=True):
        self.file = TFRecordFile(filepath, verify_crc)
        self.batch_size, self.num_samples, self.reduced = int(batch_size), int(num_samples), bool(reduced)
        self.shuffle, self.repeat, self.audio_max_length = bool(shuffle), bool(repeat), int(audio_max_length)
        if self.num_samples > self.audio_max_length:
            raise ValueError("num_samples %d > audio_max_length %d (tf.slice would fail, nsynth.py:31)"
                             % (self.num_samples, self.audio_max_length))
        self._rng = np.random.default_rng(seed)
        self._n = len(self.file)
        self._cursor = 0                 # next record to enter the shuffle buffer
        self._buffer = []                # tf.data shuffle(buffer_size=10000), nsynth.py:41
        self._done = self._n == 0

    def _draw(self):
        """One record index in dataset order (shuffle buffer, then repeat)."""
        if self.shuffle:
            while len(self._buffer) < 10000 and self._fill_one():
                pass
            if not self._buffer:
                return None
            j = int(self._rng.integers(len(self._buffer)))
            self._buffer[j], self._buffer[-1] = self._buffer[-1], self._buffer[j]
            return self._buffer.pop()
        return self._next_sequential()

    def _next_sequential(self):
        if self._cursor >= self._n:
            if not self.repeat or self._n == 0:
                return None
            self._cursor = 0
        i = self._cursor
        self._cursor += 1
        return i

    def _fill_one(self):
        # shuffle sits BEFORE repeat in the pipeline: the buffer drains at the end of an epoch, then refills
        if self._cursor >= self._n:
            return False
        self._buffer.append(self._cursor)
        self._cursor += 1
        return True

    def _indices(self):
        out = []
        while len(out) < self.batch_size:
            i = self._draw()
            if i is None:
                if self.shuffle and self.repeat and self._n:   # epoch drained: start the next one
                    self._cursor = 0
                    continue
                break
            out.append(i)
        return out

    def next(self):
        idx = self._indices()
        if not idx:
            raise StopIteration("end of %s (the reference raises tf.errors.OutOfRangeError)" % self.file.path)
        if self.reduced:
            audio, pitch = self.file.batch(idx, self.num_samples, self.audio_max_length, "pitch")
            if pitch.min() < 0 or pitch.max() > 127:
                onehot = np.zeros((len(idx), 128), np.float32)      # tf.one_hot: out-of-range -> all zeros
                ok = (pitch >= 0) & (pitch < 128)
                onehot[np.nonzero(ok)[0], pitch[ok]] = 1.0
            else:
                onehot = np.eye(128, dtype=np.float32)[pitch]
            return audio, onehot                                     # nsynth.py:27-33
        out: Dict[str, list] = {k: [] for k in FEATURES}
        for i in idx:
            for k, (kind, length) in FEATURES.items():
                v = self.file.feature(i, k)
                want = self.audio_max_length if length == "audio_max_length" else length
                if want is not None and len(v) != want:
                    raise RuntimeError("feature '%s' of record %d holds %d values, expected %d (FixedLenFeature, "
                                       "nsynth.py:10-25)" % (k, i, len(v), want))
                out[k].append(v)
        return {k: (np.array(v, dtype=object) if FEATURES[k][0] == 1 else np.stack(v)) for k, v in out.items()}

    __next__ = next

    def __iter__(self):
        return self
