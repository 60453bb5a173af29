#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "../../include/srwn_io.h"
int main(int argc, char** argv) {
  const char* path = argv[1];
  FILE* f = fopen(path, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> raw(n); fread(raw.data(), 1, n, f); fclose(f);
  unsigned long long len0; memcpy(&len0, raw.data(), 8);
  srand(1);
  int ok = 0, err = 0;
  for (int trial = 0; trial < 3000; ++trial) {
    std::vector<unsigned char> b = raw;
    int m = 1 + rand() % 8;
    for (int i = 0; i < m; ++i) b[12 + rand() % len0] = (unsigned char)(rand() & 255);
    if (trial % 7 == 0) b.resize(12 + rand() % (b.size() - 12));      // truncation
    FILE* o = fopen("/tmp/srwn_fuzz.tfrecord", "wb"); fwrite(b.data(), 1, b.size(), o); fclose(o);
    void* h = srwn_tfr_open("/tmp/srwn_fuzz.tfrecord", trial % 3 == 0);
    if (!h) { ++err; continue; }
    const char* keys[] = {"audio", "pitch", "note_str", "qualities", "nope"};
    for (const char* k : keys) {
      int kind; long long cnt;
      if (srwn_tfr_feature(h, 0, k, &kind, (int64_t*)&cnt) == 0) ++ok; else ++err;
      float fb[64]; int64_t ib[16]; char cb[32]; int64_t got;
      srwn_tfr_read_floats(h, 0, k, fb, 64, &got);
      srwn_tfr_read_int64s(h, 0, k, ib, 16, &got);
      srwn_tfr_read_bytes(h, 0, k, cb, 32, &got);
    }
    int64_t idx[2] = {0, srwn_tfr_count(h) - 1}; float audio[2 * 64]; int64_t lab[2];
    srwn_tfr_read_batch(h, idx, 2, "audio", 640, 64, audio, "pitch", lab, 2);
    srwn_tfr_close(h);
  }
  printf("ok=%d err=%d\n", ok, err);
  return 0;
}
