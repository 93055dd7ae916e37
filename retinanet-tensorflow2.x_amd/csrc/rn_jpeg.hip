// rn_jpeg.hip — host-side JPEG decoder for the TFRecord input path: COCO's TFRecords hold JPEG bytes
// and the reference decodes them with tf.io.decode_image(channels=3) (dataloader/tfrecord_parser.py:20-23), i.e.
// libjpeg(-turbo) with its defaults: JDCT_ISLOW inverse DCT and "fancy" chroma up-sampling.  No codec library is
// available to this build, so the decoder is restated here from the JPEG standard (ITU-T T.81) and from libjpeg's
// published integer algorithms, so that pixels come out as libjpeg produces them:
//   * baseline / extended sequential Huffman (SOF0 / SOF1, 8-bit), 1 (grayscale -> replicated to RGB) or 3
//     components (YCbCr), any sampling factors up to 2x2 for luma with 1x1 chroma (4:4:4, 4:2:2, 4:4:0, 4:2:0),
//     restart intervals, APPn / COM skipped; progressive (SOF2), arithmetic coding, 12-bit and CMYK are rejected
//     with RN_EINVAL and a message (tf would decode progressive ones; none of them is on the hot path of the bench);
//   * inverse DCT: jidctint.c's "islow" LL&M integer transform (13-bit constants, PASS1_BITS 2), same descales;
//   * chroma up-sampling: jdsample.c's h2v1 / h2v2 "fancy" triangle filters ((3a+b+1|2)>>2, (3*this+near+8|7)>>4),
//     v2-only by replication of the h1v2 fancy rule, box replication otherwise;
//   * YCbCr -> RGB: jdcolor.c's 16-bit fixed-point tables (FIX(1.40200) etc., ONE_HALF rounding).
// PARITY UNPINNED against libjpeg itself (not installed): tests pin the entropy decoder exactly through an independent
// encoder, the IDCT against a float64 IDCT (<= 1 level) and whole images against a float pipeline
// (tests/test_jpeg_cpu.py).  Host only: nothing here touches the GPU.
#include <stdint.h>
#include <string.h>

#include <exception>
#include <vector>

#include "rn_common.h"

namespace {

struct Huff {
  // canonical Huffman table: for code length l (1..16): mincode, maxcode (-1 = none), valptr
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool present = false;
};

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int bw = 0, bh = 0;            // blocks per row / rows (padded to whole MCUs)
  int dw = 0, dh = 0;            // downsampled size in samples (ceil)
  std::vector<uint8_t> plane;    // bw*8 x bh*8 samples
  int pred = 0;
};

struct Bits {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int n = 0;
  bool hit_marker = false;
  inline void fill() {
    while (n <= 24) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p++;
        if (b == 0xFF) {
          const int m = p < end ? *p : 0;
          if (m == 0) {
            ++p;              // stuffed zero
          } else {
            hit_marker = true; // leave the marker for the caller; feed zeros
            --p;
            b = 0;
          }
        }
      }
      acc |= (uint32_t)b << (24 - n);
      n += 8;
    }
  }
  inline int get(int k) {   // k in 0..16
    if (k == 0) return 0;
    if (n < k) fill();
    const int v = (int)(acc >> (32 - k));
    acc <<= k;
    n -= k;
    return v;
  }
  inline void reset() { acc = 0; n = 0; hit_marker = false; }
};

int build_huff(Huff& h, const uint8_t* counts, const uint8_t* vals, int nvals) {
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    k += counts[l - 1];
    code += counts[l - 1];
    h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  if (k != nvals || nvals > 256) return -1;
  memcpy(h.vals, vals, nvals);
  h.present = true;
  return 0;
}

inline int decode_symbol(Bits& b, const Huff& h) {
  int code = b.get(1);
  for (int l = 1; l <= 16; ++l) {
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    code = (code << 1) | b.get(1);
  }
  return -1;
}

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// jidctint.c (JDCT_ISLOW): CONST_BITS 13, PASS1_BITS 2.  The arithmetic runs on a wrapping 32-bit integer: on a valid
// file no intermediate leaves the int range and the results are libjpeg's bit for bit; on hostile coefficients (a 16-bit
// DQT entry times a 16-bit coefficient) the sums wrap instead of being signed-overflow / negative-left-shift undefined
// behaviour (found by the sanitizer fuzz run, tests/test_jpeg_cpu.py).
struct wi {
  uint32_t v;
  wi() : v(0) {}
  wi(int x) : v((uint32_t)x) {}
  int s() const { return (int)v; }
};
inline wi operator+(wi a, wi b) { wi r; r.v = a.v + b.v; return r; }
inline wi operator-(wi a, wi b) { wi r; r.v = a.v - b.v; return r; }
inline wi operator*(wi a, int k) { wi r; r.v = a.v * (uint32_t)k; return r; }
inline wi operator<<(wi a, int n) { wi r; r.v = a.v << n; return r; }
inline wi& operator+=(wi& a, wi b) { a.v += b.v; return a; }
inline wi& operator*=(wi& a, int k) { a.v *= (uint32_t)k; return a; }
inline int descale(wi x, int n) { return (int)(x.v + (1u << (n - 1))) >> n; }
inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void idct_islow(const int* in /* dequantized, natural order */, uint8_t* out, int stride) {
  constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
                F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  int ws[64];
  for (int c = 0; c < 8; ++c) {   // pass 1: columns
    const int* p = in + c;
    wi z2 = p[16], z3 = p[48];
    wi z1 = (z2 + z3) * F0541;
    wi tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = p[0]; z3 = p[32];
    wi tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
    const wi tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    wi z4 = tmp1 + tmp3;
    const wi z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    int* w = ws + c;
    w[0] = descale(tmp10 + tmp3, 11);  w[56] = descale(tmp10 - tmp3, 11);
    w[8] = descale(tmp11 + tmp2, 11);  w[48] = descale(tmp11 - tmp2, 11);
    w[16] = descale(tmp12 + tmp1, 11); w[40] = descale(tmp12 - tmp1, 11);
    w[24] = descale(tmp13 + tmp0, 11); w[32] = descale(tmp13 - tmp0, 11);
  }
  for (int r = 0; r < 8; ++r) {   // pass 2: rows
    const int* p = ws + r * 8;
    wi z2 = p[2], z3 = p[6];
    wi z1 = (z2 + z3) * F0541;
    wi tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    wi tmp0 = (wi(p[0]) + wi(p[4])) << 13, tmp1 = (wi(p[0]) - wi(p[4])) << 13;
    const wi tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = p[7]; tmp1 = p[5]; tmp2 = p[3]; tmp3 = p[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    wi z4 = tmp1 + tmp3;
    const wi z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    uint8_t* o = out + r * stride;
    // (the level shift is added to a value that may sit at the edge of the int range on hostile data: saturate first)
    auto px = [](wi x) { const int d = descale(x, 18); return clamp255((d > 1024 ? 1024 : (d < -1024 ? -1024 : d)) + 128); };
    o[0] = px(tmp10 + tmp3); o[7] = px(tmp10 - tmp3);
    o[1] = px(tmp11 + tmp2); o[6] = px(tmp11 - tmp2);
    o[2] = px(tmp12 + tmp1); o[5] = px(tmp12 - tmp1);
    o[3] = px(tmp13 + tmp0); o[4] = px(tmp13 - tmp0);
  }
}

struct Decoder {
  const uint8_t* data;
  size_t len;
  int width = 0, height = 0, ncomp = 0, restart = 0;
  uint16_t qt[4][64];
  bool qt_ok[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  Comp comp[3];
  int hmax = 1, vmax = 1;
  bool got_sof = false;
};

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

int parse_headers(Decoder& d, size_t& scan_pos, bool header_only) {
  const uint8_t* p = d.data;
  if (d.len < 4 || p[0] != 0xFF || p[1] != 0xD8) {
    rn_set_error("jpeg: no SOI marker");
    return RN_EINVAL;
  }
  size_t pos = 2;
  while (pos + 4 <= d.len) {
    if (p[pos] != 0xFF) {
      rn_set_error("jpeg: marker expected at byte %zu", pos);
      return RN_EINVAL;
    }
    while (pos < d.len && p[pos] == 0xFF) ++pos;   // fill bytes
    if (pos >= d.len) break;
    const int m = p[pos++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) break;
    if (pos + 2 > d.len) break;
    const int seg = be16(p + pos);
    if (seg < 2 || pos + seg > d.len) {
      rn_set_error("jpeg: truncated segment (marker 0x%02X)", m);
      return RN_EINVAL;
    }
    const uint8_t* s = p + pos + 2;
    const int n = seg - 2;
    if (m == 0xC0 || m == 0xC1) {
      if (n < 6 || s[0] != 8) {
        rn_set_error("jpeg: only 8-bit precision is supported");
        return n < 6 ? RN_EINVAL : RN_EUNSUPPORTED;
      }
      d.height = be16(s + 1); d.width = be16(s + 3); d.ncomp = s[5];
      if (d.width <= 0 || d.height <= 0 || d.ncomp < 1 || n < 6 + 3 * d.ncomp) {   // corrupt: not a job for another decoder
        rn_set_error("jpeg: bad frame header (%d x %d, %d components)", d.width, d.height, d.ncomp);
        return RN_EINVAL;
      }
      if (d.ncomp != 1 && d.ncomp != 3) {   // CMYK / YCCK
        rn_set_error("jpeg: unsupported frame (%d x %d, %d components)", d.width, d.height, d.ncomp);
        return RN_EUNSUPPORTED;
      }
      if ((long long)d.width * d.height > (64ll << 20)) {   // untrusted bytes: a corrupt header must not ask for gigabytes
        rn_set_error("jpeg: frame %d x %d exceeds the 64 Mpixel limit", d.width, d.height);
        return RN_EINVAL;
      }
      d.hmax = d.vmax = 1;          // a repeated SOF starts over
      for (int i = 0; i < d.ncomp; ++i) {
        Comp& c = d.comp[i];
        c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i] & 3;
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2) {
          rn_set_error("jpeg: sampling factor %dx%d of component %d is not supported", c.h, c.v, i);
          return (c.h < 1 || c.v < 1) ? RN_EINVAL : RN_EUNSUPPORTED;
        }
        d.hmax = c.h > d.hmax ? c.h : d.hmax;
        d.vmax = c.v > d.vmax ? c.v : d.vmax;
      }
      if (d.ncomp == 3 && (d.comp[1].h != 1 || d.comp[1].v != 1 || d.comp[2].h != 1 || d.comp[2].v != 1)) {
        rn_set_error("jpeg: chroma components must be sampled 1x1");
        return RN_EUNSUPPORTED;
      }
      if (d.ncomp == 1) { d.comp[0].h = d.comp[0].v = 1; d.hmax = d.vmax = 1; }   // a single component is never interleaved
      d.got_sof = true;
      if (header_only) return RN_OK;
    } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      rn_set_error("jpeg: SOF%d (progressive / lossless / arithmetic) is not supported, only baseline Huffman", m - 0xC0);
      return RN_EUNSUPPORTED;
    } else if (m == 0xDB) {
      int q = 0;
      while (q < n) {
        const int pq = s[q] >> 4, tq = s[q] & 15;
        if (tq > 3 || (pq != 0 && pq != 1) || q + 1 + 64 * (pq + 1) > n) {
          rn_set_error("jpeg: bad DQT");
          return RN_EINVAL;
        }
        for (int i = 0; i < 64; ++i) d.qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(s + q + 1 + 2 * i) : s[q + 1 + i];
        d.qt_ok[tq] = true;
        q += 1 + 64 * (pq + 1);
      }
    } else if (m == 0xC4) {
      int q = 0;
      while (q + 17 <= n) {
        const int tc = s[q] >> 4, th = s[q] & 15;
        int total = 0;
        for (int i = 0; i < 16; ++i) total += s[q + 1 + i];
        if (tc > 1 || th > 3 || q + 17 + total > n || build_huff(tc ? d.ac[th] : d.dc[th], s + q + 1, s + q + 17, total)) {
          rn_set_error("jpeg: bad DHT");
          return RN_EINVAL;
        }
        q += 17 + total;
      }
    } else if (m == 0xDD) {
      if (n >= 2) d.restart = be16(s);
    } else if (m == 0xDA) {
      if (!d.got_sof || n < 1 || n < 1 + 2 * s[0] + 3 || s[0] != d.ncomp) {
        rn_set_error("jpeg: SOS before SOF, or a non-interleaved multi-scan file (not supported)");
        return d.got_sof ? RN_EUNSUPPORTED : RN_EINVAL;
      }
      for (int i = 0; i < d.ncomp; ++i) {
        int ci = -1;
        for (int j = 0; j < d.ncomp; ++j)
          if (d.comp[j].id == s[1 + 2 * i]) ci = j;
        if (ci != i) {
          rn_set_error("jpeg: scan component order differs from the frame's");
          return RN_EINVAL;
        }
        d.comp[i].td = s[2 + 2 * i] >> 4; d.comp[i].ta = s[2 + 2 * i] & 15;
        if (d.comp[i].td > 3 || d.comp[i].ta > 3 || !d.dc[d.comp[i].td].present || !d.ac[d.comp[i].ta].present ||
            !d.qt_ok[d.comp[i].tq]) {
          rn_set_error("jpeg: scan refers to a missing table");
          return RN_EINVAL;
        }
      }
      scan_pos = pos + seg;
      return RN_OK;
    }
    pos += seg;
  }
  rn_set_error(d.got_sof ? "jpeg: no scan found" : "jpeg: no frame header found");
  return RN_EINVAL;
}

int decode_scan(Decoder& d, size_t scan_pos) {
  const int mcux = (d.width + 8 * d.hmax - 1) / (8 * d.hmax), mcuy = (d.height + 8 * d.vmax - 1) / (8 * d.vmax);
  for (int i = 0; i < d.ncomp; ++i) {
    Comp& c = d.comp[i];
    c.bw = mcux * c.h; c.bh = mcuy * c.v;
    c.dw = (d.width * c.h + d.hmax - 1) / d.hmax; c.dh = (d.height * c.v + d.vmax - 1) / d.vmax;
    c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
    c.pred = 0;
  }
  Bits b;
  b.p = d.data + scan_pos;
  b.end = d.data + d.len;
  int coef[64];
  int until_restart = d.restart;
  for (int my = 0; my < mcuy; ++my) {
    for (int mx = 0; mx < mcux; ++mx) {
      if (d.restart && until_restart == 0) {
        // byte-align, expect RSTn
        b.reset();
        const uint8_t* q = b.p;
        while (q + 1 < b.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
        if (q + 1 >= b.end) {
          rn_set_error("jpeg: restart marker missing");
          return RN_EINVAL;
        }
        b.p = q + 2;
        for (int i = 0; i < d.ncomp; ++i) d.comp[i].pred = 0;
        until_restart = d.restart;
      }
      for (int i = 0; i < d.ncomp; ++i) {
        Comp& c = d.comp[i];
        const uint16_t* q = d.qt[c.tq];
        for (int by = 0; by < c.v; ++by) {
          for (int bx = 0; bx < c.h; ++bx) {
            memset(coef, 0, sizeof(coef));
            const int t = decode_symbol(b, d.dc[c.td]);
            if (t < 0 || t > 11) {
              rn_set_error("jpeg: corrupt DC code");
              return RN_EINVAL;
            }
            c.pred += t ? extend(b.get(t), t) : 0;
            // a valid stream keeps DC differences inside 12 bits: clamp so that corrupt data cannot overflow the products
            c.pred = c.pred < -32768 ? -32768 : (c.pred > 32767 ? 32767 : c.pred);
            coef[0] = c.pred * q[0];
            for (int k = 1; k < 64;) {
              const int rs = decode_symbol(b, d.ac[c.ta]);
              if (rs < 0) {
                rn_set_error("jpeg: corrupt AC code");
                return RN_EINVAL;
              }
              const int r = rs >> 4, s = rs & 15;
              if (s == 0) {
                if (r != 15) break;   // EOB
                k += 16;
                continue;
              }
              k += r;
              if (k > 63) {
                rn_set_error("jpeg: AC coefficient index out of range");
                return RN_EINVAL;
              }
              const int z = kZigzag[k];
              coef[z] = extend(b.get(s), s) * q[z];
              ++k;
            }
            uint8_t* out = c.plane.data() + ((size_t)(my * c.v + by) * 8) * (c.bw * 8) + (size_t)(mx * c.h + bx) * 8;
            idct_islow(coef, out, c.bw * 8);
          }
        }
      }
      if (d.restart) --until_restart;
    }
  }
  return RN_OK;
}

// jdsample.c: full-size plane [height][width] of a component sampled (h, v) against (hmax, vmax)
void upsample(const Decoder& d, const Comp& c, std::vector<uint8_t>& out) {
  const int W = d.width, H = d.height, sw = c.bw * 8;
  out.assign((size_t)W * H, 0);
  const int hx = d.hmax / c.h, vx = d.vmax / c.v;
  const uint8_t* src = c.plane.data();
  if (hx == 1 && vx == 1) {
    for (int y = 0; y < H; ++y) memcpy(out.data() + (size_t)y * W, src + (size_t)y * sw, W);
    return;
  }
  const int dw = c.dw, dh = c.dh;
  std::vector<int> colsum(dw);
  for (int y = 0; y < H; ++y) {
    const int sy = vx == 2 ? y >> 1 : y;
    const uint8_t* r0 = src + (size_t)sy * sw;
    uint8_t* o = out.data() + (size_t)y * W;
    if (vx == 2) {
      // nearer row = sy, further row = above for even output rows, below for odd ones; edges replicate
      int oy = (y & 1) ? sy + 1 : sy - 1;
      oy = oy < 0 ? 0 : (oy > dh - 1 ? dh - 1 : oy);
      const uint8_t* r1 = src + (size_t)oy * sw;
      for (int x = 0; x < dw; ++x) colsum[x] = 3 * r0[x] + r1[x];
      if (hx == 2) {   // h2v2_fancy_upsample
        for (int x = 0; x < dw; ++x) {
          const int cur = colsum[x], last = x > 0 ? colsum[x - 1] : 0, next = x + 1 < dw ? colsum[x + 1] : 0;
          const int e = x == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + last + 8) >> 4;
          const int f = x == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + next + 7) >> 4;
          if (2 * x < W) o[2 * x] = (uint8_t)e;
          if (2 * x + 1 < W) o[2 * x + 1] = (uint8_t)f;
        }
      } else {         // h1v2 fancy: (3 * near + far + 1 | 2) >> 2, rounding alternates by output row
        const int bias = (y & 1) ? 2 : 1;
        for (int x = 0; x < W; ++x) o[x] = (uint8_t)((colsum[x] + bias) >> 2);
      }
    } else {           // h2v1_fancy_upsample
      for (int x = 0; x < dw; ++x) {
        const int cur = r0[x];
        const int e = x == 0 ? cur : (3 * cur + r0[x - 1] + 1) >> 2;
        const int f = x == dw - 1 ? cur : (3 * cur + r0[x + 1] + 2) >> 2;
        if (2 * x < W) o[2 * x] = (uint8_t)e;
        if (2 * x + 1 < W) o[2 * x + 1] = (uint8_t)f;
      }
    }
  }
}

}  // namespace

// width / height / components (1 or 3) of a baseline JPEG; RN_EINVAL with a message for anything else
extern "C" int rn_jpeg_info(const void* data, size_t len, int32_t* width, int32_t* height, int32_t* components) try {
  RN_CHECK_ARG(data && width && height && components, "rn_jpeg_info: null argument");
  Decoder d;
  d.data = (const uint8_t*)data;
  d.len = len;
  size_t scan = 0;
  const int rc = parse_headers(d, scan, true);
  if (rc) return rc;
  *width = d.width; *height = d.height; *components = d.ncomp;
  return RN_OK;
} catch (const std::exception& e) {      // no C++ exception crosses the C boundary (std::bad_alloc on a hostile header)
  rn_set_error("rn_jpeg_info: %s", e.what());
  return RN_ENOMEM;
}

// RGB u8 [height, width, 3] (grayscale replicated), decoded like libjpeg's defaults (see the header of this file)
extern "C" int rn_jpeg_decode(const void* data, size_t len, uint8_t* rgb_out, size_t out_bytes) try {
  RN_CHECK_ARG(data && rgb_out, "rn_jpeg_decode: null argument");
  Decoder d;
  d.data = (const uint8_t*)data;
  d.len = len;
  size_t scan = 0;
  int rc = parse_headers(d, scan, false);
  if (rc) return rc;
  RN_CHECK_ARG(out_bytes >= (size_t)d.width * d.height * 3, "rn_jpeg_decode: output buffer too small");
  rc = decode_scan(d, scan);
  if (rc) return rc;
  const int W = d.width, H = d.height;
  std::vector<uint8_t> y, cb, cr;
  upsample(d, d.comp[0], y);
  if (d.ncomp == 1) {
    for (size_t i = 0; i < (size_t)W * H; ++i) rgb_out[3 * i] = rgb_out[3 * i + 1] = rgb_out[3 * i + 2] = y[i];
    return RN_OK;
  }
  upsample(d, d.comp[1], cb);
  upsample(d, d.comp[2], cr);
  // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, FIX(x) = (int)(x * 65536 + 0.5)
  int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) {
    const int x = i - 128;
    cr_r[i] = (91881 * x + 32768) >> 16;      // FIX(1.40200)
    cb_b[i] = (116130 * x + 32768) >> 16;     // FIX(1.77200)
    cr_g[i] = -46802 * x;                     // FIX(0.71414)
    cb_g[i] = -22554 * x + 32768;             // FIX(0.34414), ONE_HALF folded in
  }
  for (size_t i = 0; i < (size_t)W * H; ++i) {
    const int yy = y[i], b = cb[i], r = cr[i];
    rgb_out[3 * i] = clamp255(yy + cr_r[r]);
    rgb_out[3 * i + 1] = clamp255(yy + ((cb_g[b] + cr_g[r]) >> 16));
    rgb_out[3 * i + 2] = clamp255(yy + cb_b[b]);
  }
  return RN_OK;
} catch (const std::exception& e) {
  rn_set_error("rn_jpeg_decode: %s", e.what());
  return RN_ENOMEM;
}

// the inverse DCT on its own, for the tests (coefficients already dequantized, natural order)
extern "C" int rn_jpeg_idct_islow(const int32_t* coef64, uint8_t* out64) {
  RN_CHECK_ARG(coef64 && out64, "rn_jpeg_idct_islow: null argument");
  int c[64];
  for (int i = 0; i < 64; ++i) c[i] = coef64[i];
  idct_islow(c, out64, 8);
  return RN_OK;
}
