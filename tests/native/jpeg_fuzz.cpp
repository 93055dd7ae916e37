// Sanitizer fuzz harness for the native JPEG decoder (csrc/rn_jpeg.hip, host code): built by tests/test_jpeg_cpu.py with
// hipcc --cuda-host-only -fsanitize=address,undefined and run on mutated copies of a few seed files.  Any out-of-bounds
// read / write, signed overflow or uncaught exception aborts the process; decode errors are the expected outcome.
//   jpeg_fuzz <mutations per seed> <seed.jpg> [<seed.jpg> ...]
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

extern "C" int rn_jpeg_info(const uint8_t* data, size_t n, int32_t* w, int32_t* h, int32_t* c);
extern "C" int rn_jpeg_decode(const uint8_t* data, size_t n, uint8_t* out, size_t out_bytes);
void rn_set_error(const char*, ...) {}   // rn_core.hip's error slot is not linked into the harness

static uint64_t g_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
  g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(g_state >> 33);
}

static int decode(const std::vector<uint8_t>& buf) {
  // exact-size heap copy: one byte past the end is an ASan error
  uint8_t* p = (uint8_t*)malloc(buf.size() ? buf.size() : 1);
  memcpy(p, buf.data(), buf.size());
  int32_t w = 0, h = 0, c = 0;
  int ok = 0;
  if (rn_jpeg_info(p, buf.size(), &w, &h, &c) == 0 && w > 0 && h > 0 && (long long)w * h <= (1 << 22)) {
    std::vector<uint8_t> out((size_t)w * h * 3);
    ok = rn_jpeg_decode(p, buf.size(), out.data(), out.size()) == 0;
  }
  free(p);
  return ok;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int per_seed = atoi(argv[1]);
  long decoded = 0, total = 0;
  for (int a = 2; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb");
    if (!f) return 3;
    std::vector<uint8_t> seed;
    uint8_t tmp[4096];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) seed.insert(seed.end(), tmp, tmp + n);
    fclose(f);
    if (!decode(seed)) return 4;   // the seeds themselves must decode
    for (int it = 0; it < per_seed; ++it) {
      std::vector<uint8_t> m = seed;
      switch (rnd() % 6) {
        case 0:   // a few byte flips anywhere
          for (uint32_t k = 1 + rnd() % 4; k > 0; --k) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
          break;
        case 1:   // random bytes in the header region (tables, frame, scan headers)
          for (uint32_t k = 1 + rnd() % 6; k > 0; --k) m[rnd() % (m.size() < 700 ? m.size() : 700)] = (uint8_t)rnd();
          break;
        case 2:   // truncation
          m.resize(rnd() % m.size());
          break;
        case 3: { // a segment length field set to an extreme
          size_t i = 2;
          const uint32_t pick = rnd() % 8;
          for (uint32_t s = 0; i + 4 <= m.size() && m[i] == 0xff; ++s) {
            const size_t len = ((size_t)m[i + 2] << 8) | m[i + 3];
            if (s == pick) {
              const uint32_t v = rnd() % 4;
              const uint16_t nl = v == 0 ? 0 : v == 1 ? 2 : v == 2 ? 0xffff : (uint16_t)rnd();
              m[i + 2] = (uint8_t)(nl >> 8); m[i + 3] = (uint8_t)nl;
              break;
            }
            if (m[i + 1] == 0xda) break;
            i += 2 + len;
          }
          break;
        }
        case 4:   // 0xff / marker bytes injected into the entropy-coded data
          for (uint32_t k = 1 + rnd() % 3; k > 0; --k) {
            const size_t at = m.size() / 2 + rnd() % (m.size() / 2);
            m[at] = 0xff;
            if (at + 1 < m.size()) m[at + 1] = (uint8_t)(0xc0 + rnd() % 0x3f);
          }
          break;
        default:  // a chunk cut out of the middle
          if (m.size() > 64) {
            const size_t at = rnd() % (m.size() - 32), len = 1 + rnd() % 31;
            m.erase(m.begin() + at, m.begin() + at + len);
          }
          break;
      }
      decoded += decode(m);
      ++total;
    }
  }
  printf("%ld mutated inputs, %ld still decoded\n", total, decoded);
  return 0;
}
