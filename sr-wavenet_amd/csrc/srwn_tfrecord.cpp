// TFRecord + tf.train.Example reader for the NSynth files the reference trains on (nsynth.py:6-46,
// filter_tfrecord.py:40-58), host-only C++ (no TensorFlow, no protobuf library):
//   record  = u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)      (little endian)
//   payload = Example{ features(1): Features{ feature(1): map<string, Feature> } },
//             Feature = oneof { bytes_list(1){bytes value(1)}, float_list(2){float value(1)}, int64_list(3){int64 value(1)} }
//             (repeated scalars are accepted packed or unpacked)
// The file is mapped once and indexed (offset of every record); batches are decoded by a small thread pool straight
// into the caller's arrays.  C-ABI in include/srwn_io.h.
#include <mutex>
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/srwn_io.h"

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slice-by-8 tables
uint32_t g_tab[8][256];
std::once_flag g_tab_once;
void crc_build() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
    g_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_tab[t][i] = (g_tab[t - 1][i] >> 8) ^ g_tab[0][g_tab[t - 1][i] & 0xff];
}
void crc_init() { std::call_once(g_tab_once, crc_build); }   // first callers may arrive on several threads
uint32_t crc32c(const uint8_t* p, size_t n) {
  uint32_t c = 0xffffffffu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = g_tab[7][lo & 0xff] ^ g_tab[6][(lo >> 8) & 0xff] ^ g_tab[5][(lo >> 16) & 0xff] ^ g_tab[4][lo >> 24] ^
        g_tab[3][hi & 0xff] ^ g_tab[2][(hi >> 8) & 0xff] ^ g_tab[1][(hi >> 16) & 0xff] ^ g_tab[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}
inline uint32_t masked(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

struct File {
  int fd = -1;
  const uint8_t* base = nullptr;
  size_t size = 0;
  std::vector<uint64_t> off;   // payload offsets
  std::vector<uint64_t> len;   // payload lengths
};

// ---- minimal protobuf wire reader ---------------------------------------------------------------
struct Span { const uint8_t* p; const uint8_t* e; };
bool varint(Span& s, uint64_t& v) {
  v = 0;
  for (int sh = 0; sh < 64 && s.p < s.e; sh += 7) {
    const uint8_t b = *s.p++;
    v |= (uint64_t)(b & 0x7f) << sh;
    if (!(b & 0x80)) return true;
  }
  return false;
}
bool skip(Span& s, uint32_t wt) {
  uint64_t v;
  switch (wt) {
    case 0: return varint(s, v);
    case 1: if (s.e - s.p < 8) return false; s.p += 8; return true;
    case 2: if (!varint(s, v) || (uint64_t)(s.e - s.p) < v) return false; s.p += v; return true;
    case 5: if (s.e - s.p < 4) return false; s.p += 4; return true;
    default: return false;
  }
}
bool sub(Span& s, Span& out) {   // length-delimited field body
  uint64_t n;
  if (!varint(s, n) || (uint64_t)(s.e - s.p) < n) return false;
  out = {s.p, s.p + n};
  s.p += n;
  return true;
}

// finds Feature `key` in an Example payload; returns its (kind, list body) -- kind 1 bytes, 2 float, 3 int64.
// A key that occurs more than once resolves to its LAST entry (protobuf map semantics: what tf.parse_single_example sees).
int find_feature(Span ex, const char* key, size_t klen, int& kind, Span& list) {
  uint64_t tag;
  bool found = false;
  while (ex.p < ex.e) {
    if (!varint(ex, tag)) return SRWN_IO_E_PARSE;
    if (tag == ((1u << 3) | 2)) {   // Example.features
      Span feats;
      if (!sub(ex, feats)) return SRWN_IO_E_PARSE;
      while (feats.p < feats.e) {
        if (!varint(feats, tag)) return SRWN_IO_E_PARSE;
        if (tag != ((1u << 3) | 2)) { if (!skip(feats, tag & 7)) return SRWN_IO_E_PARSE; continue; }
        Span entry;                  // map entry { key = 1, value = 2 }
        if (!sub(feats, entry)) return SRWN_IO_E_PARSE;
        Span k{nullptr, nullptr}, val{nullptr, nullptr};
        while (entry.p < entry.e) {
          if (!varint(entry, tag)) return SRWN_IO_E_PARSE;
          if (tag == ((1u << 3) | 2)) { if (!sub(entry, k)) return SRWN_IO_E_PARSE; }
          else if (tag == ((2u << 3) | 2)) { if (!sub(entry, val)) return SRWN_IO_E_PARSE; }
          else if (!skip(entry, tag & 7)) return SRWN_IO_E_PARSE;
        }
        if (!k.p || (size_t)(k.e - k.p) != klen || memcmp(k.p, key, klen) != 0) continue;
        if (!val.p) return SRWN_IO_E_PARSE;
        kind = 0;
        list = {val.p, val.p};       // an empty Feature is an empty list
        while (val.p < val.e) {
          if (!varint(val, tag)) return SRWN_IO_E_PARSE;
          const uint32_t f = (uint32_t)(tag >> 3);
          if ((tag & 7) == 2 && f >= 1 && f <= 3) { kind = (int)f; if (!sub(val, list)) return SRWN_IO_E_PARSE; }
          else if (!skip(val, tag & 7)) return SRWN_IO_E_PARSE;
        }
        found = true;                // keep scanning: a later entry with the same key replaces this one
      }
    } else if (!skip(ex, tag & 7)) {
      return SRWN_IO_E_PARSE;
    }
  }
  return found ? 0 : SRWN_IO_E_NOKEY;
}

int read_floats(Span list, float* out, int64_t max_n, int64_t& n) {
  n = 0;
  uint64_t tag;
  while (list.p < list.e) {
    if (!varint(list, tag) || (tag >> 3) != 1) return SRWN_IO_E_PARSE;
    if ((tag & 7) == 2) {            // packed
      Span body;
      if (!sub(list, body) || (body.e - body.p) % 4) return SRWN_IO_E_PARSE;
      const int64_t cnt = (body.e - body.p) / 4;
      const int64_t take = std::max<int64_t>(0, std::min(cnt, max_n - n));
      if (out && take) memcpy(out + n, body.p, (size_t)take * 4);
      n += cnt;
    } else if ((tag & 7) == 5) {
      if (list.e - list.p < 4) return SRWN_IO_E_PARSE;
      if (out && n < max_n) memcpy(out + n, list.p, 4);
      list.p += 4;
      ++n;
    } else {
      return SRWN_IO_E_PARSE;
    }
  }
  return 0;
}

int read_int64s(Span list, int64_t* out, int64_t max_n, int64_t& n) {
  n = 0;
  uint64_t tag, v;
  while (list.p < list.e) {
    if (!varint(list, tag) || (tag >> 3) != 1) return SRWN_IO_E_PARSE;
    if ((tag & 7) == 2) {
      Span body;
      if (!sub(list, body)) return SRWN_IO_E_PARSE;
      while (body.p < body.e) {
        if (!varint(body, v)) return SRWN_IO_E_PARSE;
        if (out && n < max_n) out[n] = (int64_t)v;
        ++n;
      }
    } else if ((tag & 7) == 0) {
      if (!varint(list, v)) return SRWN_IO_E_PARSE;
      if (out && n < max_n) out[n] = (int64_t)v;
      ++n;
    } else {
      return SRWN_IO_E_PARSE;
    }
  }
  return 0;
}

File* as_file(void* h) { return reinterpret_cast<File*>(h); }
int check_idx(File* f, int64_t i) {
  if (!f) return fail(SRWN_IO_E_ARG, "null handle");
  if (i < 0 || (size_t)i >= f->off.size()) return fail(SRWN_IO_E_ARG, "record %lld out of range [0, %zu)", (long long)i, f->off.size());
  return 0;
}

}  // namespace

extern "C" const char* srwn_io_last_error(void) { return g_err; }

// CRC-32C (Castagnoli) of a host buffer, and TensorFlow's masking of it (records, table blocks and tensor bundle
// entries all store the masked value)
extern "C" uint32_t srwn_crc32c(const void* data, uint64_t n) {
  crc_init();
  return crc32c(static_cast<const uint8_t*>(data), (size_t)n);
}
extern "C" uint32_t srwn_crc32c_mask(uint32_t crc) { return masked(crc); }

extern "C" void* srwn_tfr_open(const char* path, int32_t verify_crc) {
  crc_init();
  if (!path) { fail(SRWN_IO_E_ARG, "null path"); return nullptr; }
  File* f = new File;
  f->fd = open(path, O_RDONLY);
  struct stat st;
  if (f->fd < 0 || fstat(f->fd, &st) != 0) {
    fail(SRWN_IO_E_IO, "cannot open %s", path);
    if (f->fd >= 0) close(f->fd);
    delete f;
    return nullptr;
  }
  f->size = (size_t)st.st_size;
  if (f->size) {
    void* m = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, f->fd, 0);
    if (m == MAP_FAILED) { fail(SRWN_IO_E_IO, "mmap failed for %s", path); close(f->fd); delete f; return nullptr; }
    f->base = reinterpret_cast<const uint8_t*>(m);
  }
  size_t pos = 0;
  while (pos < f->size) {
    if (f->size - pos < 12) { fail(SRWN_IO_E_FORMAT, "truncated record header at byte %zu", pos); goto bad; }
    {
      uint64_t len;
      uint32_t c;
      memcpy(&len, f->base + pos, 8);
      memcpy(&c, f->base + pos + 8, 4);
      if (verify_crc && masked(crc32c(f->base + pos, 8)) != c) { fail(SRWN_IO_E_CRC, "length CRC mismatch at byte %zu", pos); goto bad; }
      if (len > f->size - pos - 12 || f->size - pos - 12 - len < 4) { fail(SRWN_IO_E_FORMAT, "truncated record payload at byte %zu", pos); goto bad; }
      if (verify_crc) {
        memcpy(&c, f->base + pos + 12 + len, 4);
        if (masked(crc32c(f->base + pos + 12, (size_t)len)) != c) { fail(SRWN_IO_E_CRC, "payload CRC mismatch in record %zu", f->off.size()); goto bad; }
      }
      f->off.push_back(pos + 12);
      f->len.push_back(len);
      pos += 12 + len + 4;
    }
  }
  return f;
bad:
  srwn_tfr_close(f);
  return nullptr;
}

extern "C" void srwn_tfr_close(void* h) {
  File* f = as_file(h);
  if (!f) return;
  if (f->base) munmap(const_cast<uint8_t*>(f->base), f->size);
  if (f->fd >= 0) close(f->fd);
  delete f;
}

extern "C" int64_t srwn_tfr_count(void* h) { return h ? (int64_t)as_file(h)->off.size() : -1; }

extern "C" int srwn_tfr_feature(void* h, int64_t idx, const char* key, int32_t* kind, int64_t* count) {
  File* f = as_file(h);
  if (int rc = check_idx(f, idx)) return rc;
  if (!key || !kind || !count) return fail(SRWN_IO_E_ARG, "null argument");
  Span ex{f->base + f->off[idx], f->base + f->off[idx] + f->len[idx]}, list;
  int k = 0;
  int rc = find_feature(ex, key, strlen(key), k, list);
  if (rc) return fail(rc, rc == SRWN_IO_E_NOKEY ? "feature '%s' not in record %lld" : "malformed Example ('%s', record %lld)", key, (long long)idx);
  *kind = k;
  int64_t n = 0;
  if (k == 2) rc = read_floats(list, nullptr, 0, n);
  else if (k == 3) rc = read_int64s(list, nullptr, 0, n);
  else if (k == 1) {
    uint64_t tag;
    Span b;
    while (list.p < list.e) { if (!varint(list, tag) || tag != ((1u << 3) | 2) || !sub(list, b)) { rc = SRWN_IO_E_PARSE; break; } ++n; }
  }
  if (rc) return fail(rc, "malformed list in feature '%s'", key);
  *count = n;
  return 0;
}

extern "C" int srwn_tfr_read_floats(void* h, int64_t idx, const char* key, float* out, int64_t max_n, int64_t* n_out) {
  File* f = as_file(h);
  if (int rc = check_idx(f, idx)) return rc;
  if (!key || !n_out || (max_n > 0 && !out)) return fail(SRWN_IO_E_ARG, "null argument");
  Span ex{f->base + f->off[idx], f->base + f->off[idx] + f->len[idx]}, list;
  int kind = 0;
  int rc = find_feature(ex, key, strlen(key), kind, list);
  if (rc) return fail(rc, rc == SRWN_IO_E_NOKEY ? "feature '%s' not in record %lld" : "malformed Example ('%s', record %lld)", key, (long long)idx);
  if (kind != 2 && list.p != list.e) return fail(SRWN_IO_E_TYPE, "feature '%s' is not a float_list", key);
  rc = read_floats(list, out, max_n, *n_out);
  return rc ? fail(rc, "malformed float_list '%s'", key) : 0;
}

extern "C" int srwn_tfr_read_int64s(void* h, int64_t idx, const char* key, int64_t* out, int64_t max_n, int64_t* n_out) {
  File* f = as_file(h);
  if (int rc = check_idx(f, idx)) return rc;
  if (!key || !n_out || (max_n > 0 && !out)) return fail(SRWN_IO_E_ARG, "null argument");
  Span ex{f->base + f->off[idx], f->base + f->off[idx] + f->len[idx]}, list;
  int kind = 0;
  int rc = find_feature(ex, key, strlen(key), kind, list);
  if (rc) return fail(rc, rc == SRWN_IO_E_NOKEY ? "feature '%s' not in record %lld" : "malformed Example ('%s', record %lld)", key, (long long)idx);
  if (kind != 3 && list.p != list.e) return fail(SRWN_IO_E_TYPE, "feature '%s' is not an int64_list", key);
  rc = read_int64s(list, out, max_n, *n_out);
  return rc ? fail(rc, "malformed int64_list '%s'", key) : 0;
}

extern "C" int srwn_tfr_read_bytes(void* h, int64_t idx, const char* key, char* out, int64_t max_n, int64_t* n_out) {
  File* f = as_file(h);
  if (int rc = check_idx(f, idx)) return rc;
  if (!key || !n_out || (max_n > 0 && !out)) return fail(SRWN_IO_E_ARG, "null argument");
  Span ex{f->base + f->off[idx], f->base + f->off[idx] + f->len[idx]}, list;
  int kind = 0;
  int rc = find_feature(ex, key, strlen(key), kind, list);
  if (rc) return fail(rc, rc == SRWN_IO_E_NOKEY ? "feature '%s' not in record %lld" : "malformed Example ('%s', record %lld)", key, (long long)idx);
  if (kind != 1 && list.p != list.e) return fail(SRWN_IO_E_TYPE, "feature '%s' is not a bytes_list", key);
  *n_out = 0;
  uint64_t tag;
  Span b;
  if (list.p < list.e) {   // first value (the NSynth string features hold exactly one)
    if (!varint(list, tag) || tag != ((1u << 3) | 2) || !sub(list, b)) return fail(SRWN_IO_E_PARSE, "malformed bytes_list '%s'", key);
    *n_out = b.e - b.p;
    if (out) memcpy(out, b.p, (size_t)std::min<int64_t>(*n_out, max_n));
  }
  return 0;
}

// audio [B, num_samples] <- the first num_samples floats of `audio_key` (which must hold exactly audio_len floats when
// audio_len > 0: tf.FixedLenFeature([audio_max_length]), nsynth.py:15);  label [B] <- first int64 of `label_key`.
extern "C" int srwn_tfr_read_batch(void* h, const int64_t* idx, int32_t B, const char* audio_key, int64_t audio_len,
                                   int32_t num_samples, float* audio, const char* label_key, int64_t* label,
                                   int32_t nthreads) {
  File* f = as_file(h);
  if (!f || !idx || !audio_key || !audio || B < 0 || num_samples < 0) return fail(SRWN_IO_E_ARG, "bad argument");
  for (int i = 0; i < B; ++i)
    if (int rc = check_idx(f, idx[i])) return rc;
  if (audio_len > 0 && num_samples > audio_len) return fail(SRWN_IO_E_ARG, "num_samples %d > audio length %lld", num_samples, (long long)audio_len);
  nthreads = std::max(1, std::min<int>(nthreads, B));
  std::vector<int> rcs(nthreads, 0);
  std::vector<std::string> msgs(nthreads);
  auto work = [&](int t) {
    for (int i = t; i < B; i += nthreads) {
      int64_t n = 0;
      int rc = srwn_tfr_read_floats(h, idx[i], audio_key, audio + (size_t)i * num_samples, num_samples, &n);
      if (!rc && audio_len > 0 && n != audio_len)
        rc = fail(SRWN_IO_E_SHAPE, "feature '%s' of record %lld holds %lld floats, expected %lld", audio_key, (long long)idx[i], (long long)n, (long long)audio_len);
      if (!rc && n < num_samples)
        rc = fail(SRWN_IO_E_SHAPE, "feature '%s' of record %lld holds %lld floats, need %d", audio_key, (long long)idx[i], (long long)n, num_samples);
      if (!rc && label_key && label) {
        int64_t m = 0;
        rc = srwn_tfr_read_int64s(h, idx[i], label_key, label + i, 1, &m);
        if (!rc && m < 1) rc = fail(SRWN_IO_E_SHAPE, "feature '%s' of record %lld is empty", label_key, (long long)idx[i]);
      }
      if (rc) { rcs[t] = rc; msgs[t] = g_err; return; }
    }
  };
  if (nthreads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < nthreads; ++t)
    if (rcs[t]) return fail(rcs[t], "%s", msgs[t].c_str());
  return 0;
}
