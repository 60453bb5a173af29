"""NSynth TFRecord reader (include/srwn_io.h, sr-wavenet_amd/nsynth.py) vs an independent pure-Python writer.

The reference reads these files through TensorFlow (nsynth.py:9-45) and ships no fixture; the wire formats are public:
TFRecord framing with masked CRC-32C, tf.train.Example protobuf.  The writer below encodes them by hand (its CRC is
pinned by the standard CRC-32C check value), the C++ reader must give back exactly what was written."""
import ctypes
import os
import re
import struct

import numpy as np
import pytest

from tests._pkg import sub

_TAB = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _TAB.append(_c)


def crc32c(b: bytes) -> int:
    c = 0xFFFFFFFF
    for x in b:
        c = _TAB[(c ^ x) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def ld(field: int, body: bytes) -> bytes:
    return varint((field << 3) | 2) + varint(len(body)) + body


def feature(value, packed=True) -> bytes:
    if isinstance(value, bytes):
        return ld(1, ld(1, value))
    a = np.asarray(value)
    if a.dtype.kind == "f":
        a = a.astype("<f4")
        body = ld(1, a.tobytes()) if packed else b"".join(varint((1 << 3) | 5) + struct.pack("<f", x) for x in a)
        return ld(2, body)
    body = ld(1, b"".join(varint(int(x)) for x in a)) if packed else b"".join(varint(1 << 3) + varint(int(x)) for x in a)
    return ld(3, body)


def example(feats: dict, packed=True) -> bytes:
    entries = b"".join(ld(1, ld(1, k.encode()) + ld(2, feature(v, packed))) for k, v in feats.items())
    return ld(1, entries)


def record(payload: bytes) -> bytes:
    hdr = struct.pack("<Q", len(payload))
    return hdr + struct.pack("<I", masked(crc32c(hdr))) + payload + struct.pack("<I", masked(crc32c(payload)))


def nsynth_example(i: int, audio_len: int, rng) -> dict:
    return {
        "sample_rate": [4000], "note_str": b"bass_synthetic_%03d-060-100" % i, "qualities": rng.integers(0, 2, 10),
        "audio": rng.standard_normal(audio_len).astype(np.float32), "instrument_family": [i % 11], "pitch": [21 + i % 88],
        "instrument_source": [i % 3], "instrument_str": b"bass_synthetic_%03d" % i,
        "instrument_source_str": b"synthetic", "note": [100000 + i], "instrument": [i], "instrument_family_str": b"bass",
        "velocity": [25 * (1 + i % 5)],
    }


@pytest.fixture
def nsynth_file(tmp_path):
    rng = np.random.default_rng(0)
    exs = [nsynth_example(i, 640, rng) for i in range(23)]
    path = tmp_path / "synthetic.tfrecord"
    with open(path, "wb") as f:
        for i, e in enumerate(exs):
            f.write(record(example(e, packed=(i % 2 == 0))))     # alternate packed / unpacked repeated scalars
    return str(path), exs


def test_crc32c_known_answers():
    assert crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value
    assert crc32c(b"\x00" * 32) == 0x8A9136AA                    # RFC 3720 B.4 test vector
    assert masked(crc32c(b"123456789")) == ((0xE3069283 >> 15 | 0xE3069283 << 17) + 0xA282EAD8) & 0xFFFFFFFF


def test_io_library_exports_header():
    NS = sub("nsynth")
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(NS.__file__)), "include", "srwn_io.h")).read()
    names = set(re.findall(r"\b(srwn_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(NS.LIB_PATH)
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), n


def test_every_feature_round_trips(nsynth_file):
    NS = sub("nsynth")
    path, exs = nsynth_file
    f = NS.TFRecordFile(path)
    assert len(f) == len(exs)
    for i, e in enumerate(exs):
        for k, v in e.items():
            got = f.feature(i, k)
            if isinstance(v, bytes):
                assert got == v, (i, k)
            else:
                assert np.array_equal(got, np.asarray(v)), (i, k)
                assert got.dtype == (np.float32 if np.asarray(v).dtype.kind == "f" else np.int64)
    with pytest.raises(RuntimeError, match="not in record"):
        f.feature(0, "qualities_str")                               # commented out in the reference too (nsynth.py:12)
    with pytest.raises(RuntimeError, match="out of range"):
        f.feature(len(exs), "pitch")
    a, p = f.batch([3, 3, 22, 0], 100, audio_len=640)
    assert np.array_equal(a[0], exs[3]["audio"][:100]) and np.array_equal(a[2], exs[22]["audio"][:100])
    assert p.tolist() == [exs[3]["pitch"][0], exs[3]["pitch"][0], exs[22]["pitch"][0], exs[0]["pitch"][0]]
    with pytest.raises(RuntimeError, match="expected 64000"):
        f.batch([0], 100, audio_len=64000)                          # FixedLenFeature([audio_max_length]) mismatch
    f.close()


def test_reader_reduced_mode_epochs_and_batches(nsynth_file):
    NS = sub("nsynth")
    path, exs = nsynth_file
    by_pitch_audio = {e["audio"][:50].tobytes(): e["pitch"][0] for e in exs}
    # no shuffle, no repeat: file order, short last batch, then the end
    r = NS.NsynthDataReader(path, 5, num_samples=50, shuffle=False, repeat=False, audio_max_length=640)
    sizes, seen = [], []
    while True:
        try:
            a, y = r.next()
        except StopIteration:
            break
        assert a.dtype == np.float32 and y.shape == (len(a), 128) and y.dtype == np.float32
        assert np.array_equal(y.argmax(1), [by_pitch_audio[x.tobytes()] for x in a]) and (y.sum(1) == 1).all()
        sizes.append(len(a)); seen += [x.tobytes() for x in a]
    assert sizes == [5, 5, 5, 5, 3] and seen == [e["audio"][:50].tobytes() for e in exs]
    # shuffle + repeat: every epoch is a permutation of the file; batches run across the epoch boundary
    r = NS.NsynthDataReader(path, 4, num_samples=50, shuffle=True, repeat=True, audio_max_length=640, seed=1)
    stream = []
    for _ in range(3 * 23 // 4 + 1):
        a, _y = r.next()
        assert len(a) == 4
        stream += [x.tobytes() for x in a]
    want = sorted(e["audio"][:50].tobytes() for e in exs)
    assert sorted(stream[:23]) == want and sorted(stream[23:46]) == want and stream[:23] != stream[23:46]
    assert stream[:23] != [e["audio"][:50].tobytes() for e in exs]
    with pytest.raises(ValueError):
        NS.NsynthDataReader(path, 4, num_samples=1000, audio_max_length=640)


def test_reader_full_mode_matches_parse_single_example(nsynth_file):
    NS = sub("nsynth")
    path, exs = nsynth_file
    r = NS.NsynthDataReader(path, 1, num_samples=640, reduced=False, shuffle=False, repeat=False, audio_max_length=640)
    d = r.next()                                                    # filter_tfrecord.py:31-38 indexes d[key][0]
    assert set(d) == set(exs[0])
    assert d["pitch"][0][0] == exs[0]["pitch"][0] and d["audio"].shape == (1, 640) and d["qualities"].shape == (1, 10)
    assert d["note_str"][0] == exs[0]["note_str"] and d["sample_rate"][0][0] == 4000
    r2 = NS.NsynthDataReader(path, 2, num_samples=64, reduced=False, shuffle=False, repeat=False, audio_max_length=64000)
    with pytest.raises(RuntimeError, match="expected 64000"):
        r2.next()


def test_corruption_is_detected(nsynth_file, tmp_path):
    NS = sub("nsynth")
    path, _ = nsynth_file
    raw = bytearray(open(path, "rb").read())
    bad = tmp_path / "bad.tfrecord"
    flipped = bytearray(raw); flipped[200] ^= 0x40
    bad.write_bytes(flipped)
    with pytest.raises(RuntimeError, match="CRC"):
        NS.TFRecordFile(str(bad))
    assert len(NS.TFRecordFile(str(bad), verify_crc=False)) == 23   # framing intact, payload damaged
    bad.write_bytes(raw[:-7])
    with pytest.raises(RuntimeError, match="truncated"):
        NS.TFRecordFile(str(bad))
    hdr = bytearray(raw); hdr[3] ^= 0x01                            # length field of the first record
    bad.write_bytes(hdr)
    with pytest.raises(RuntimeError, match="CRC"):
        NS.TFRecordFile(str(bad))
    with pytest.raises(RuntimeError, match="cannot open"):
        NS.TFRecordFile(str(tmp_path / "missing.tfrecord"))
    empty = tmp_path / "empty.tfrecord"; empty.write_bytes(b"")
    r = NS.NsynthDataReader(str(empty), 4, num_samples=10, audio_max_length=10)
    with pytest.raises(StopIteration):
        r.next()
    garbage = tmp_path / "garbage.tfrecord"; garbage.write_bytes(record(b"\xff\xff\xff\xff\xff"))
    g = NS.TFRecordFile(str(garbage))
    with pytest.raises(RuntimeError, match="malformed"):
        g.feature(0, "audio")


def test_simple_audio_wave_batches():
    """generator.py's synthetic data source (simple_audio.py:40-66): shapes, range, label convention, wave shapes."""
    SA = sub("simple_audio")
    rs = np.random.RandomState(0)
    x, y = SA.generate_wave_batch(6, 5120, rng=rs)
    assert x.shape == (6, 5120) and y.shape == (6, 10)
    assert np.allclose(x.min(1), -1) and np.allclose(x.max(1), 1)
    assert (y.sum(1) == 1).all() and set(np.unique(y)) == {0.0, 1.0}
    t = SA.CreateTicks(1, 1000)
    assert len(t) == 1000 and t[0] == 0 and t[-1] == 1
    s = SA.Sine(5, 1, 1000); q = SA.Square(5, 1, 1000); w = SA.Sawtooth(5, 1, 1000); tr = SA.Triangle(5, 1, 1000)
    for v in (s, q, w, tr):
        assert v.shape == (1000,) and v.min() >= -1 - 1e-12 and v.max() <= 1 + 1e-12
    assert abs(s[50] - np.sin(2 * np.pi * 5 * t[50])) < 1e-12
    assert set(np.unique(q)) == {-1.0, 1.0} and q[10] == 1 and q[150] == -1          # first half-period high
    assert w[1] > w[0] and abs(w[0] + 1) < 1e-12                                     # ramps up from -1
    assert abs(tr[0] + 1) < 1e-12 and abs(tr.max() - 1) < 0.05 and tr[50] > tr[0]    # -1 -> 1 -> -1 per period
    # label = one-hot of int(f/2 - 1) - 10 for f in 22..39  ->  indices 0..8
    assert y.argmax(1).min() >= 0 and y.argmax(1).max() <= 8
    assert np.allclose(SA.Normalize(np.array([2.0, 4.0, 3.0]), -1, 1), [-1, 1, 0])


def test_parser_survives_mutated_payloads(nsynth_file, tmp_path):
    """Robustness of the protobuf wire reader: random byte damage inside payloads (CRC checks off) must end in a
    value or a RuntimeError, never in a crash or an out-of-bounds read (every length is checked against its span)."""
    NS = sub("nsynth")
    path, exs = nsynth_file
    raw = open(path, "rb").read()
    rng = np.random.default_rng(0)
    bad = tmp_path / "fuzz.tfrecord"
    first_len = struct.unpack("<Q", raw[:8])[0]
    outcomes = {"ok": 0, "err": 0}
    for trial in range(150):
        buf = bytearray(raw)
        for _ in range(int(rng.integers(1, 6))):
            pos = 12 + int(rng.integers(0, first_len))           # inside the first record's payload
            buf[pos] = int(rng.integers(0, 256))
        bad.write_bytes(buf)
        f = NS.TFRecordFile(str(bad), verify_crc=False)
        for key in ("audio", "pitch", "note_str", "qualities"):
            try:
                v = f.feature(0, key)
                assert v is not None
                outcomes["ok"] += 1
            except RuntimeError:
                outcomes["err"] += 1
        try:
            f.batch([0, 1], 64, audio_len=640)
        except RuntimeError:
            pass
        f.close()
    assert outcomes["ok"] > 0 and outcomes["err"] > 0
    # truncated varints / lengths running past the record
    for payload in (b"\x0a\xff\xff\xff\xff\x0f", b"\x0a\x05\x0a\x03\x0a\x01", b"\x0a\x80", b"\x0a\x02\x0a\x7f"):
        bad.write_bytes(record(payload))
        g = NS.TFRecordFile(str(bad))
        with pytest.raises(RuntimeError):
            g.feature(0, "audio")
        g.close()


def test_duplicate_feature_key_resolves_to_the_last_entry(tmp_path):
    """protobuf map semantics (what tf.parse_single_example sees, nsynth.py:27): a key written twice keeps its LAST
    value; and the CRC table is built once however many threads arrive first."""
    import threading
    NS = sub("nsynth")
    entry = lambda k, v: ld(1, ld(1, k.encode()) + ld(2, feature(v)))
    ex = ld(1, entry("pitch", [10]) + entry("audio", np.arange(4, dtype=np.float32)) + entry("pitch", [77]))
    path = tmp_path / "dup.tfrecord"
    with open(path, "wb") as f:
        f.write(record(ex))
    r = NS.TFRecordFile(str(path))
    assert r.feature(0, "pitch").tolist() == [77]
    assert r.feature(0, "audio").tolist() == [0.0, 1.0, 2.0, 3.0]
    r.close()
    out = []
    def work():
        f2 = NS.TFRecordFile(str(path)); out.append(f2.feature(0, "pitch").tolist()); f2.close()
    ts = [threading.Thread(target=work) for _ in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert out == [[77]] * 8
    # a trailing duplicate with an EMPTY Feature body is still the last entry: it replaces the value (no kind, no
    # elements), it does not leave the earlier one in place
    ex2 = ld(1, entry("pitch", [10]) + ld(1, ld(1, b"pitch") + ld(2, b"")) + entry("audio", np.arange(2, dtype=np.float32)))
    path2 = tmp_path / "dup_empty.tfrecord"
    with open(path2, "wb") as f:
        f.write(record(ex2))
    r2 = NS.TFRecordFile(str(path2))
    assert len(r2.feature(0, "pitch")) == 0
    assert r2.feature(0, "audio").tolist() == [0.0, 1.0]
    r2.close()
