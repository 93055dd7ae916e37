// rn_tfrecord.hip — HOST side of SURVEY §8(f)-4: TFRecord framing (CRC-32C) and the tf.train.Example wire
// format for the reference's feature set.  No device code; lives in librnet_hip.so so the Python input pipeline
// (retinanet/dataloader/{input_pipeline,tfrecord_parser}.py, retinanet/dataset_utils/tfrecord_writer.py) has
// the same native reader TensorFlow gives the reference (tf.data.TFRecordDataset + parse_single_example,
// input_pipeline.py:60-68, tfrecord_parser.py:4-41).
//
// Formats restated from their public definitions (TensorFlow is not vendored in the reference):
//   record   = u64le len | u32le mask(crc32c(len)) | payload[len] | u32le mask(crc32c(payload))
//   mask(c)  = ((c >> 15) | (c << 17)) + 0xa282ead8
//   Example  = { 1: Features }            Features = { 1: repeated MapEntry{1: string key, 2: Feature} }
//   Feature  = oneof { 1: BytesList, 2: FloatList, 3: Int64List }, each list = { 1: repeated value }
//              (float / int64 values packed by proto3 writers, unpacked accepted)
#include <string.h>

#include "rn_common.h"

namespace {

// ---- CRC-32C ---------------------------------------------------------------------------------------
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
const CrcTables g_crc;

uint32_t crc32c_sw(uint32_t crc, const uint8_t* p, size_t n) {   // slicing-by-8
  while (n && ((uintptr_t)p & 7)) { crc = (crc >> 8) ^ g_crc.t[0][(crc ^ *p++) & 0xff]; --n; }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= crc;
    crc = g_crc.t[7][v & 0xff] ^ g_crc.t[6][(v >> 8) & 0xff] ^ g_crc.t[5][(v >> 16) & 0xff] ^
          g_crc.t[4][(v >> 24) & 0xff] ^ g_crc.t[3][(v >> 32) & 0xff] ^ g_crc.t[2][(v >> 40) & 0xff] ^
          g_crc.t[1][(v >> 48) & 0xff] ^ g_crc.t[0][(v >> 56) & 0xff];
    p += 8; n -= 8;
  }
  while (n--) crc = (crc >> 8) ^ g_crc.t[0][(crc ^ *p++) & 0xff];
  return crc;
}

inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }
inline uint32_t load_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t load_u64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// ---- protobuf wire reader --------------------------------------------------------------------------
struct Span {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 70 && p < end; shift += 7) {
      const uint8_t b = *p++;
      if (shift < 64) v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  Span bytes() {   // length-delimited field body
    const uint64_t n = varint();
    if (!ok || n > (uint64_t)(end - p)) { ok = false; return Span{end, end, false}; }
    Span s{p, p + n};
    p += n;
    return s;
  }
  void skip(int wire) {
    switch (wire) {
      case 0: varint(); break;
      case 1: if (end - p < 8) ok = false; else p += 8; break;
      case 2: bytes(); break;
      case 5: if (end - p < 4) ok = false; else p += 4; break;
      default: ok = false;
    }
  }
};

enum Kind { K_NONE = 0, K_BYTES = 1, K_FLOAT = 2, K_INT64 = 3 };

struct FeatureView {   // the (last) list of one Feature; an empty Feature (no oneof set) has kind NONE
  int kind = K_NONE;
  Span list{nullptr, nullptr};
};

bool read_feature(Span f, FeatureView* out) {
  out->kind = K_NONE;
  while (!f.done() && f.ok) {
    const uint64_t tag = f.varint();
    const int field = (int)(tag >> 3), wire = (int)(tag & 7);
    if (field >= 1 && field <= 3 && wire == 2) {
      out->kind = field;   // oneof: the last one set wins
      out->list = f.bytes();
    } else {
      f.skip(wire);
    }
  }
  return f.ok && out->list.ok;
}

// BytesList: count values and remember the first
bool read_bytes_list(Span l, int* count, Span* first) {
  *count = 0;
  while (!l.done() && l.ok) {
    const uint64_t tag = l.varint();
    if ((tag >> 3) == 1 && (tag & 7) == 2) {
      Span v = l.bytes();
      if (*count == 0) *first = v;
      ++*count;
    } else {
      l.skip((int)(tag & 7));
    }
  }
  return l.ok;
}

bool read_float_list(Span l, float* dst, int cap, int* count) {
  int n = 0;
  while (!l.done() && l.ok) {
    const uint64_t tag = l.varint();
    const int field = (int)(tag >> 3), wire = (int)(tag & 7);
    if (field == 1 && wire == 2) {           // packed
      Span v = l.bytes();
      if (!v.ok || ((v.end - v.p) & 3)) return false;
      for (; v.p < v.end; v.p += 4, ++n)
        if (n < cap) memcpy(dst + n, v.p, 4);
    } else if (field == 1 && wire == 5) {    // unpacked
      if (l.end - l.p < 4) return false;
      if (n < cap) memcpy(dst + n, l.p, 4);
      l.p += 4; ++n;
    } else {
      l.skip(wire);
    }
  }
  *count = n;
  return l.ok;
}

bool read_int64_list(Span l, int64_t* dst, int cap, int* count) {
  int n = 0;
  while (!l.done() && l.ok) {
    const uint64_t tag = l.varint();
    const int field = (int)(tag >> 3), wire = (int)(tag & 7);
    if (field == 1 && wire == 2) {
      Span v = l.bytes();
      if (!v.ok) return false;
      while (!v.done() && v.ok) {
        const int64_t x = (int64_t)v.varint();
        if (n < cap) dst[n] = x;
        ++n;
      }
      if (!v.ok) return false;
    } else if (field == 1 && wire == 0) {
      const int64_t x = (int64_t)l.varint();
      if (n < cap) dst[n] = x;
      ++n;
    } else {
      l.skip(wire);
    }
  }
  *count = n;
  return l.ok;
}

// ---- protobuf wire writer --------------------------------------------------------------------------
struct Writer {
  uint8_t* out;
  size_t cap, n = 0;
  void byte(uint8_t b) { if (out && n < cap) out[n] = b; ++n; }
  void varint(uint64_t v) { while (v >= 0x80) { byte((uint8_t)(v | 0x80)); v >>= 7; } byte((uint8_t)v); }
  void raw(const void* p, size_t k) { if (out && n + k <= cap) memcpy(out + n, p, k); n += k; }
};
size_t varint_size(uint64_t v) { size_t k = 1; while (v >= 0x80) { v >>= 7; ++k; } return k; }

}  // namespace

extern "C" uint32_t rn_crc32c(const void* data, size_t nbytes) {
  return ~crc32c_sw(0xffffffffu, (const uint8_t*)data, nbytes);
}

extern "C" uint32_t rn_crc32c_masked(const void* data, size_t nbytes) { return mask_crc(rn_crc32c(data, nbytes)); }

extern "C" long long rn_tfrecord_scan(const uint8_t* buf, size_t nbytes, uint64_t* payload_offsets,
                                      uint64_t* payload_lengths, long long max_records, int verify_crc,
                                      int allow_partial_tail, size_t* consumed) {
  RN_CHECK_ARG(buf || nbytes == 0, "rn_tfrecord_scan: null buffer");
  size_t pos = 0;
  long long n = 0;
  while (pos < nbytes && n < max_records) {
    const size_t left = nbytes - pos;
    if (left < 12) {
      if (allow_partial_tail) break;
      rn_set_error("rn_tfrecord_scan: truncated record header at byte %zu", pos);
      return RN_EINVAL;
    }
    const uint64_t len = load_u64(buf + pos);
    if (verify_crc && load_u32(buf + pos + 8) != rn_crc32c_masked(buf + pos, 8)) {
      rn_set_error("rn_tfrecord_scan: corrupted record length at byte %zu", pos);
      return RN_EINVAL;
    }
    if (len > left - 12 || left - 12 - len < 4) {
      if (allow_partial_tail) break;
      rn_set_error("rn_tfrecord_scan: truncated record at byte %zu (length %llu)", pos, (unsigned long long)len);
      return RN_EINVAL;
    }
    if (verify_crc && load_u32(buf + pos + 12 + len) != rn_crc32c_masked(buf + pos + 12, len)) {
      rn_set_error("rn_tfrecord_scan: corrupted record data at byte %zu", pos);
      return RN_EINVAL;
    }
    if (payload_offsets) payload_offsets[n] = pos + 12;
    if (payload_lengths) payload_lengths[n] = len;
    ++n;
    pos += 16 + len;
  }
  if (consumed) *consumed = pos;
  return n;
}

extern "C" size_t rn_tfrecord_frame(const uint8_t* payload, size_t nbytes, uint8_t* out) {
  const uint64_t len = nbytes;
  memcpy(out, &len, 8);
  const uint32_t c0 = rn_crc32c_masked(out, 8);
  memcpy(out + 8, &c0, 4);
  if (nbytes) memcpy(out + 12, payload, nbytes);
  const uint32_t c1 = rn_crc32c_masked(out + 12, nbytes);
  memcpy(out + 12 + nbytes, &c1, 4);
  return nbytes + 16;
}

extern "C" int rn_example_parse(const uint8_t* record, size_t nbytes, rn_example_info* info, float* xmins,
                                float* ymins, float* xmaxs, float* ymaxs, int64_t* classes, int capacity) {
  RN_CHECK_ARG(record && info, "rn_example_parse: null argument");
  RN_CHECK_ARG(capacity >= 0, "rn_example_parse: negative capacity");
  memset(info, 0, sizeof(*info));
  // the map: key -> last Feature seen (protobuf map semantics)
  static const char* const kKeys[7] = {"image", "image_id", "xmins", "ymins", "xmaxs", "ymaxs", "classes"};
  FeatureView fv[7];
  bool seen[7] = {false, false, false, false, false, false, false};
  Span ex{record, record + nbytes};
  while (!ex.done() && ex.ok) {
    const uint64_t tag = ex.varint();
    if ((tag >> 3) == 1 && (tag & 7) == 2) {          // Example.features
      Span feats = ex.bytes();
      while (!feats.done() && feats.ok) {
        const uint64_t t2 = feats.varint();
        if ((t2 >> 3) == 1 && (t2 & 7) == 2) {        // map entry
          Span entry = feats.bytes();
          Span key{nullptr, nullptr}, val{nullptr, nullptr};
          bool has_val = false;
          while (!entry.done() && entry.ok) {
            const uint64_t t3 = entry.varint();
            if ((t3 >> 3) == 1 && (t3 & 7) == 2) key = entry.bytes();
            else if ((t3 >> 3) == 2 && (t3 & 7) == 2) { val = entry.bytes(); has_val = true; }
            else entry.skip((int)(t3 & 7));
          }
          RN_CHECK_ARG(entry.ok && key.ok && val.ok, "rn_example_parse: malformed map entry");
          const size_t klen = key.p ? (size_t)(key.end - key.p) : 0;
          for (int k = 0; k < 7; ++k)
            if (strlen(kKeys[k]) == klen && (klen == 0 || memcmp(kKeys[k], key.p, klen) == 0)) {
              FeatureView v;
              if (has_val) RN_CHECK_ARG(read_feature(val, &v), "rn_example_parse: malformed Feature '%s'", kKeys[k]);
              fv[k] = v;
              seen[k] = true;
            }
        } else {
          feats.skip((int)(t2 & 7));
        }
      }
      RN_CHECK_ARG(feats.ok, "rn_example_parse: malformed Features message");
    } else {
      ex.skip((int)(tag & 7));
    }
  }
  RN_CHECK_ARG(ex.ok, "rn_example_parse: malformed Example message");

  // FixedLenFeature([], string) 'image'
  RN_CHECK_ARG(seen[0] && fv[0].kind != K_NONE,
               "Feature: image (data type: string) is required but could not be found.");
  RN_CHECK_ARG(fv[0].kind == K_BYTES, "Feature: image.  Data types don't match. Expected type: string");
  {
    int cnt = 0;
    Span first{nullptr, nullptr};
    RN_CHECK_ARG(read_bytes_list(fv[0].list, &cnt, &first), "rn_example_parse: malformed BytesList 'image'");
    RN_CHECK_ARG(cnt == 1, "Key: image.  Can't parse serialized Example: expected 1 value, got %d", cnt);
    info->image_offset = (uint64_t)(first.p - record);
    info->image_length = (uint64_t)(first.end - first.p);
  }
  // FixedLenFeature([], int64) 'image_id'
  RN_CHECK_ARG(seen[1] && fv[1].kind != K_NONE,
               "Feature: image_id (data type: int64) is required but could not be found.");
  RN_CHECK_ARG(fv[1].kind == K_INT64, "Feature: image_id.  Data types don't match. Expected type: int64");
  {
    int cnt = 0;
    int64_t v = 0;
    RN_CHECK_ARG(read_int64_list(fv[1].list, &v, 1, &cnt), "rn_example_parse: malformed Int64List 'image_id'");
    RN_CHECK_ARG(cnt == 1, "Key: image_id.  Can't parse serialized Example: expected 1 value, got %d", cnt);
    info->image_id = v;
  }
  // VarLenFeature(float32) x4, VarLenFeature(int64)
  float* fdst[4] = {xmins, ymins, xmaxs, ymaxs};
  int32_t* fcnt[4] = {&info->n_xmins, &info->n_ymins, &info->n_xmaxs, &info->n_ymaxs};
  bool overflow = false;
  for (int k = 0; k < 4; ++k) {
    if (!seen[2 + k] || fv[2 + k].kind == K_NONE) continue;
    RN_CHECK_ARG(fv[2 + k].kind == K_FLOAT, "Feature: %s.  Data types don't match. Expected type: float", kKeys[2 + k]);
    int cnt = 0;
    RN_CHECK_ARG(read_float_list(fv[2 + k].list, fdst[k], fdst[k] ? capacity : 0, &cnt),
                 "rn_example_parse: malformed FloatList '%s'", kKeys[2 + k]);
    *fcnt[k] = cnt;
    overflow |= cnt > capacity;
  }
  if (seen[6] && fv[6].kind != K_NONE) {
    RN_CHECK_ARG(fv[6].kind == K_INT64, "Feature: classes.  Data types don't match. Expected type: int64");
    int cnt = 0;
    RN_CHECK_ARG(read_int64_list(fv[6].list, classes, classes ? capacity : 0, &cnt),
                 "rn_example_parse: malformed Int64List 'classes'");
    info->n_classes = cnt;
    overflow |= cnt > capacity;
  }
  if (overflow && capacity > 0) {
    rn_set_error("rn_example_parse: a feature list is longer than capacity %d", capacity);
    return RN_ENOMEM;
  }
  return RN_OK;
}

extern "C" size_t rn_example_serialize(const uint8_t* image, size_t image_bytes, int64_t image_id, const float* boxes,
                                       int n_boxes, const int64_t* classes, int n_classes, uint8_t* out,
                                       size_t capacity) {
  // two passes over the same emitter: pass 0 sizes the Features message, pass 1 writes
  size_t features_size = 0;
  for (int pass = 0; pass < 2; ++pass) {
    Writer w{pass ? out : nullptr, pass ? capacity : 0};
    if (pass) {
      w.byte(0x0a);                 // Example.features
      w.varint(features_size);
    }
    const size_t start = w.n;
    auto entry_header = [&](const char* key, int kind, size_t list_msg) {
      const size_t klen = strlen(key);
      const size_t feature = 1 + varint_size(list_msg) + list_msg;
      const size_t entry = 1 + varint_size(klen) + klen + 1 + varint_size(feature) + feature;
      w.byte(0x0a); w.varint(entry);
      w.byte(0x0a); w.varint(klen); w.raw(key, klen);
      w.byte(0x12); w.varint(feature);
      w.byte((uint8_t)((kind << 3) | 2)); w.varint(list_msg);
    };
    auto int64_entry = [&](const char* key, const int64_t* v, int n) {
      size_t pb = 0;
      for (int i = 0; i < n; ++i) pb += varint_size((uint64_t)v[i]);
      entry_header(key, K_INT64, n ? 1 + varint_size(pb) + pb : 0);
      if (n) {
        w.byte(0x0a); w.varint(pb);
        for (int i = 0; i < n; ++i) w.varint((uint64_t)v[i]);
      }
    };
    auto float_entry = [&](const char* key, int col) {
      const size_t pb = (size_t)n_boxes * 4;
      entry_header(key, K_FLOAT, n_boxes ? 1 + varint_size(pb) + pb : 0);
      if (n_boxes) {
        w.byte(0x0a); w.varint(pb);
        for (int i = 0; i < n_boxes; ++i) w.raw(boxes + (size_t)i * 4 + col, 4);
      }
    };
    // keys in sorted order: classes, image, image_id, xmaxs, xmins, ymaxs, ymins
    int64_entry("classes", classes, n_classes);
    entry_header("image", K_BYTES, 1 + varint_size(image_bytes) + image_bytes);
    w.byte(0x0a); w.varint(image_bytes); w.raw(image, image_bytes);
    int64_entry("image_id", &image_id, 1);
    float_entry("xmaxs", 2);
    float_entry("xmins", 0);
    float_entry("ymaxs", 3);
    float_entry("ymins", 1);
    if (!pass) features_size = w.n - start;
    else {
      if (!out) return w.n;
      return w.n <= capacity ? w.n : 0;
    }
  }
  return 0;
}
