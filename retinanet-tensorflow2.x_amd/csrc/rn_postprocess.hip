// rn_postprocess.hip — a11-a14: box decode, sigmoid, per-class top-k, per-class NMS, merge.
// Reference: retinanet/model/layers/postprocessing_ops.py:15-56 (FuseDetections), :87-117
// (TransformBoxesAndScores), :128-147 (FilterTopKDetections._filter_per_class), :434-535
// (GenerateDetections._per_class_nms) and TensorFlow's NonMaxSuppressionV5 semantics as
// restated in SURVEY.md §8(c) item 6 (TF itself is not vendored in the reference).
//
// MI355X design (not a translation of the TF op chain):
//   * The per-level head outputs [B,s,s,A*K] are already [B, n_l, K] row-major, so
//     FuseDetections is pure indexing: kernels take the 5 level pointers + anchor boundaries.
//   * compact: ONE coalesced pass over the logits computes sigmoid and appends
//     key = score_bits<<32 | ~anchor to the (image,class) candidate list iff score >
//     score_threshold.  NonMaxSuppressionV5 only ever looks at such candidates and top-k keeps
//     the k largest, so "top-k then threshold" == "threshold then top-k of the survivors":
//     the [B,5000,K] scores / [B,5000,K,4] boxes tensors of the reference are never built.
//     Sorting keys descending gives (score desc, anchor asc) = the canonical top_k order.
//   * select/sort/NMS: one 256-thread workgroup per (image,class).  Candidates stream through
//     LDS in descending order in chunks of <= 8192 keys (bitonic sort; a radix select over the
//     global list finds each chunk's lower bound when the list is longer), and a single
//     wavefront runs greedy NMS over 64 candidates at a time: IoU against the already selected
//     boxes in parallel, a 64x64 in-register suppression mask, then a 64-step ballot scan.
//   * merge: one workgroup per image sorts the K*max_det padded scores and emits the final
//     top max_det (descending, ties by class-major slot), valid count, -1 padding.
// All fp32 arithmetic is contraction-free and uses rn_math.h, so scores, keep masks and
// outputs are bit-identical to the C oracle.
#include "rn_common.h"
#include "../../include/rn_math.h"
#include <math.h>
#include <stdlib.h>

#define RN_PP_THREADS 256
#define RN_PP_MAX_LEVELS 8
#define RN_SORT_CAP 8192
#define RN_MAX_DET 256

struct PPLevels {
  int num_levels;
  const float* ptr[RN_PP_MAX_LEVELS];
  long long off[RN_PP_MAX_LEVELS + 1];
  long long vbeg[RN_PP_MAX_LEVELS + 1];
};

static int pp_fill_levels(PPLevels& lv, const float* const* ptrs, const int64_t* level_offsets, int num_levels,
                          int B, int per_anchor_items) {
  if (num_levels < 1 || num_levels > RN_PP_MAX_LEVELS || !ptrs || !level_offsets) return -1;
  lv.num_levels = num_levels;
  lv.off[0] = level_offsets[0];
  lv.vbeg[0] = 0;
  if (lv.off[0] != 0) return -1;
  for (int l = 0; l < num_levels; ++l) {
    lv.ptr[l] = ptrs[l];
    lv.off[l + 1] = level_offsets[l + 1];
    const long long n_l = lv.off[l + 1] - lv.off[l];
    if (n_l <= 0 || !ptrs[l]) return -1;
    lv.vbeg[l + 1] = lv.vbeg[l] + (long long)B * n_l * per_anchor_items;
  }
  return 0;
}

static int pp_blocks(long long items) {
  long long b = rn_cdiv(items, RN_PP_THREADS);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------
// decode  (postprocessing_ops.py:87-105)
__global__ void __launch_bounds__(RN_PP_THREADS)
decode_kernel(PPLevels lv, int B, long long A, const float4* __restrict__ anchors, float4 var, int use_var,
              float in_h, float in_w, float4* __restrict__ boxes) {
  const long long total = (long long)B * A;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / A, a = i - b * A;
    int l = 0;
    while (l + 1 < lv.num_levels && a >= lv.off[l + 1]) ++l;
    const long long n_l = lv.off[l + 1] - lv.off[l];
    float4 t = ((const float4*)lv.ptr[l])[b * n_l + (a - lv.off[l])];
    const float4 an = anchors[a];
    if (use_var) {
      t.x = t.x * var.x; t.y = t.y * var.y; t.z = t.z * var.z; t.w = t.w * var.w;
    }
    const float cx = t.x * an.z + an.x;
    const float cy = t.y * an.w + an.y;
    const float w = rn_expf(t.z) * an.z;
    const float h = rn_expf(t.w) * an.w;
    const float hw = w / 2.0f, hh = h / 2.0f;
    float4 o;
    // the reference divides [x1,y1,x2,y2] by tile(input_shape) = [h,w,h,w] (:65-69,104)
    o.x = (cx - hw) / in_h;
    o.y = (cy - hh) / in_w;
    o.z = (cx + hw) / in_h;
    o.w = (cy + hh) / in_w;
    boxes[i] = o;
  }
}

extern "C" int rn_decode_boxes(const float* const* box_preds, const int64_t* level_offsets, int num_levels,
                               int B, const float* anchors, const float* box_variance, float input_h,
                               float input_w, float* boxes, void* stream) {
  PPLevels lv;
  RN_CHECK_ARG(B > 0 && anchors && boxes, "rn_decode_boxes: bad argument");
  RN_CHECK_ARG(pp_fill_levels(lv, box_preds, level_offsets, num_levels, B, 1) == 0,
               "rn_decode_boxes: bad level table");
  const long long A = lv.off[num_levels];
  float4 var = make_float4(1.f, 1.f, 1.f, 1.f);
  if (box_variance) var = make_float4(box_variance[0], box_variance[1], box_variance[2], box_variance[3]);
  hipLaunchKernelGGL(decode_kernel, dim3(pp_blocks((long long)B * A)), dim3(RN_PP_THREADS), 0,
                     (hipStream_t)stream, lv, B, A, (const float4*)anchors, var, box_variance ? 1 : 0, input_h,
                     input_w, (float4*)boxes);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ------------------------------------------------------------------------------------------
// sigmoid scores, flattened [B,A,K]  (postprocessing_ops.py:111-114)
__global__ void __launch_bounds__(RN_PP_THREADS)
sigmoid_kernel(PPLevels lv, int B, int K, long long A, float* __restrict__ scores) {
  const long long total = lv.vbeg[lv.num_levels];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < lv.num_levels && i >= lv.vbeg[l + 1]) ++l;
    const long long local = i - lv.vbeg[l];
    const long long n_l = lv.off[l + 1] - lv.off[l];
    const long long row = local / K;
    const int k = (int)(local - row * K);
    const long long b = row / n_l, j = row - b * n_l;
    scores[(b * A + lv.off[l] + j) * K + k] = rn_sigmoidf(lv.ptr[l][local]);
  }
}

extern "C" int rn_sigmoid_scores(const float* const* class_logits, const int64_t* level_offsets,
                                 int num_levels, int B, int K, float* scores, void* stream) {
  PPLevels lv;
  RN_CHECK_ARG(B > 0 && K > 0 && scores, "rn_sigmoid_scores: bad argument");
  RN_CHECK_ARG(pp_fill_levels(lv, class_logits, level_offsets, num_levels, B, K) == 0,
               "rn_sigmoid_scores: bad level table");
  hipLaunchKernelGGL(sigmoid_kernel, dim3(pp_blocks(lv.vbeg[num_levels])), dim3(RN_PP_THREADS), 0,
                     (hipStream_t)stream, lv, B, K, lv.off[num_levels], scores);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ------------------------------------------------------------------------------------------
// candidate compaction
// float -> uint32 whose unsigned order equals the float order (negatives included)
__device__ __forceinline__ unsigned int ord_bits(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(unsigned long long key) {
  const unsigned int o = (unsigned int)(key >> 32);
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ unsigned int key_index(unsigned long long key) {
  return ~(unsigned int)(key & 0xffffffffull);
}
__device__ __forceinline__ unsigned long long make_key(float score, unsigned int idx) {
  return (((unsigned long long)ord_bits(score)) << 32) | (unsigned long long)(~idx);
}

// from logits (fused sigmoid), all levels.
// One workgroup = one chunk of 64 consecutive anchors x K classes of one (level, image) = a
// contiguous run of 64*K floats, streamed once with coalesced 16-byte loads.
//   A  every value passes a cheap logit pre-test (x < logit(thr) - margin => sigmoid(x) <= thr
//      for sure); the few survivors (a detector keeps ~0.1-5 %) are appended to an LDS list.
//   B  the dense list is walked by all 256 threads: exact rn_sigmoidf, exact threshold test,
//      rank within (workgroup, class) from an LDS counter.
//   C  ONE global atomicAdd per (workgroup, class) with a non-zero count, all issued by
//      different lanes at once, reserves the slots (a returning atomic per candidate, or per
//      wave and class, serialises on its ~us latency: 2.3 ms -> this form).
//   D  survivors write key = score_bits<<32 | ~anchor to their reserved slot.
// Sigmoid work is proportional to the number of candidates, not to the number of logits, and
// there are no 64-bit divisions in the element loop.
#define RN_CT_ANCHORS 64
#define RN_CT_GROUP 4
struct CompactTiles {
  int num_levels;
  int tile_begin[RN_PP_MAX_LEVELS + 1];  // prefix over levels of B * ceil(n_l / (RN_CT_GROUP * 64))
  int tiles_per_img[RN_PP_MAX_LEVELS];
};

__global__ void __launch_bounds__(RN_PP_THREADS)
compact_logits_kernel(PPLevels lv, CompactTiles ct, int B, int K, float thr, float x_skip,
                      int* __restrict__ counts, unsigned long long* __restrict__ keys, long long cap, int cap_list) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the list holds one whole sub-tile at worst plus a quarter: while the survivors collected so far fit that quarter
  // (a detector keeps a few per cent of its logits) the next sub-tile is appended to them
  // (cap_list = 5/4 sub-tiles, or exactly one for class counts whose larger list would not fit: set by the host)
  const int cap_sub = RN_CT_ANCHORS * K, cap_total = cap_list;
  float* l_val = (float*)smem;                                   // [cap_list] logit, then score
  unsigned short* l_meta = (unsigned short*)(l_val + cap_list);  // [cap_list] anchor_local<<8 | class
  unsigned short* l_rank = l_meta + cap_list;                    // [cap_list] rank or 0xffff
  int* c_cnt = (int*)(l_rank + cap_list + (cap_list & 1));       // [K]
  int* c_base = c_cnt + K;                                       // [K]
  int* l_n = c_base + K;                                         // [1]
  int l = 0;
  const int t = blockIdx.x;
  while (l + 1 < ct.num_levels && t >= ct.tile_begin[l + 1]) ++l;
  const int lt = t - ct.tile_begin[l];
  const int b = lt / ct.tiles_per_img[l];
  const int chunk = lt - b * ct.tiles_per_img[l];
  const int n_l = (int)(lv.off[l + 1] - lv.off[l]);
  const unsigned int anchor0 = (unsigned int)(lv.off[l] + (long long)chunk * (RN_CT_GROUP * RN_CT_ANCHORS));
  for (int i = threadIdx.x; i < K; i += RN_PP_THREADS) c_cnt[i] = 0;
  if (threadIdx.x == 0) *l_n = 0;
  __syncthreads();
  // The workgroup walks RN_CT_GROUP sub-tiles of 64 anchors; their survivors collect in the LDS list and phases B - D run
  // when the list could not take another whole sub-tile (more than a quarter of a sub-tile's logits collected), or at the
  // end — with a detector's few per cent of survivors once per workgroup: a quarter of the global atomics (1 199
  // workgroups per image used to queue on each (image, class) counter: 43 of the kernel's 124 us at batch 8) and of the
  // barriers.  (Round 4 sized the list for exactly one sub-tile, so any survivor forced a flush and only EMPTY sub-tiles
  // were merged — ADVICE r4; what round 4 measured, 124 -> 104 us, came from the four times fewer workgroups.)
  // K % 4 == 0 and at most NV 16-byte vectors per thread and sub-tile (K <= 96): the logits of sub-tile s + 1 are loaded
  // before sub-tile s is tested and flushed — a sub-tile's loads used to be issued behind the previous sub-tile's barriers
  // (three workgroups per CU, each stopping for a memory round trip per sub-tile: 1.9 TB/s over the 196 MB of a batch of 8)
  constexpr int NV = 6;
  const bool pre = (K & 3) == 0 && RN_CT_ANCHORS * K <= NV * RN_PP_THREADS * 4;
  float4 nxt[NV];
  auto load_sub = [&](int sub_) {
    const int a0_ = (chunk * RN_CT_GROUP + sub_) * RN_CT_ANCHORS;
    const int rows_ = (n_l - a0_) < RN_CT_ANCHORS ? (n_l - a0_) : RN_CT_ANCHORS;
    const float* src_ = lv.ptr[l] + ((long long)b * n_l + a0_) * K;
    const int total_ = rows_ > 0 ? rows_ * K : 0;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int i = (q * RN_PP_THREADS + (int)threadIdx.x) * 4;
      nxt[q] = i < total_ ? *(const float4*)(src_ + i) : make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    }
  };
  if (pre) load_sub(0);
  for (int sub = 0; sub < RN_CT_GROUP; ++sub) {
    const int a0 = (chunk * RN_CT_GROUP + sub) * RN_CT_ANCHORS;
    const int rows = (n_l - a0) < RN_CT_ANCHORS ? (n_l - a0) : RN_CT_ANCHORS;
    const bool last = sub == RN_CT_GROUP - 1 || a0 + RN_CT_ANCHORS >= n_l;
    if (rows > 0) {
      const float* src = lv.ptr[l] + ((long long)b * n_l + a0) * K;
      const int total = rows * K;
      const int r0 = sub * RN_CT_ANCHORS;
      // ---- A: stream + pre-test ---------------------------------------------------------------
      if (pre) {
        float4 cur[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) cur[q] = nxt[q];
        if (!last) load_sub(sub + 1);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int i = (q * RN_PP_THREADS + (int)threadIdx.x) * 4;
          if (i >= total) break;
          const float x4[4] = {cur[q].x, cur[q].y, cur[q].z, cur[q].w};
          const int r = i / K, c = i - r * K;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (!(x4[u] < x_skip)) {
              const int slot = atomicAdd(l_n, 1);
              l_val[slot] = x4[u];
              l_meta[slot] = (unsigned short)(((r0 + r) << 8) | (c + u));
            }
          }
        }
      } else if ((K & 3) == 0) {
        for (int i = threadIdx.x * 4; i < total; i += RN_PP_THREADS * 4) {
          const float4 v = *(const float4*)(src + i);
          const float x4[4] = {v.x, v.y, v.z, v.w};
          const int r = i / K, c = i - r * K;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (!(x4[u] < x_skip)) {
              const int slot = atomicAdd(l_n, 1);
              l_val[slot] = x4[u];
              l_meta[slot] = (unsigned short)(((r0 + r) << 8) | (c + u));
            }
          }
        }
      } else {
        for (int i = threadIdx.x; i < total; i += RN_PP_THREADS) {
          const float x = src[i];
          if (!(x < x_skip)) {
            const int r = i / K, c = i - r * K;
            const int slot = atomicAdd(l_n, 1);
            l_val[slot] = x;
            l_meta[slot] = (unsigned short)(((r0 + r) << 8) | c);
          }
        }
      }
    }
    __syncthreads();
    const int nl = *l_n;
    if (!last && nl + cap_sub <= cap_total) continue;   // room for another whole sub-tile
    // ---- B: exact sigmoid + threshold, rank per class ---------------------------------------------
    for (int i = threadIdx.x; i < nl; i += RN_PP_THREADS) {
      const float sc = rn_sigmoidf(l_val[i]);
      unsigned short rank = 0xffffu;
      if (sc > thr) {
        rank = (unsigned short)atomicAdd(&c_cnt[l_meta[i] & 0xff], 1);
        l_val[i] = sc;
      }
      l_rank[i] = rank;
    }
    __syncthreads();
    // ---- C: reserve global slots -------------------------------------------------------------------
    for (int c = threadIdx.x; c < K; c += RN_PP_THREADS) {
      const int n = c_cnt[c];
      c_base[c] = n ? atomicAdd(&counts[(long long)b * K + c], n) : 0;
    }
    __syncthreads();
    // ---- D: write keys -----------------------------------------------------------------------------
    for (int i = threadIdx.x; i < nl; i += RN_PP_THREADS) {
      const unsigned short rank = l_rank[i];
      if (rank != 0xffffu) {
        const int c = l_meta[i] & 0xff;
        const long long slot = (long long)c_base[c] + rank;
        if (slot < cap) keys[((long long)b * K + c) * cap + slot] = make_key(l_val[i], anchor0 + (l_meta[i] >> 8));
      }
    }
    if (last) break;
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += RN_PP_THREADS) c_cnt[i] = 0;
    if (threadIdx.x == 0) *l_n = 0;
    __syncthreads();
  }
}

// from a dense score tensor [B, n, K] (stand-alone top-k / NMS entry points)
__global__ void __launch_bounds__(RN_PP_THREADS)
compact_scores_kernel(const float* __restrict__ scores, int B, long long n, int K, float thr, int use_thr,
                      int* __restrict__ counts, unsigned long long* __restrict__ keys, long long cap) {
  const long long total = (long long)B * n * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / K;
    const int k = (int)(i - row * K);
    const long long b = row / n, j = row - b * n;
    const float s = scores[i];
    if (!use_thr || s > thr) {
      const long long list = b * K + k;
      const int slot = atomicAdd(&counts[list], 1);
      if (slot < cap) keys[list * cap + slot] = make_key(s, (unsigned int)j);
    }
  }
}

// ------------------------------------------------------------------------------------------
// block-level helpers over an LDS key array
// Sorts s[0..n) descending, n any size.  The network is the bitonic sorter of the next power of two N2 in its
// uniform-direction form — every merge stage opens with the mirrored step (t against k - 1 - t inside a block of k), then
// the half-cleaners — so every compare-exchange wants the larger key at the LOWER index: positions n .. N2 - 1, imagined to
// hold keys below every real one, never take part (a pair whose upper index is >= n is already in order) and need no
// storage.  (The alternating-direction form moved the padding through the array: it needed N2 keys of LDS — 64 KB for the
// 5 000 candidates of a soft-NMS list, two lists per compute unit where 5 056 keys allow three.)
__device__ void bitonic_sort_desc(unsigned long long* s, int n) {
  int N2 = 64;
  while (N2 < n) N2 <<= 1;
  const int half = N2 >> 1;
  for (int k = 2; k <= N2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool mirror = j == (k >> 1);
      // 4 independent compare-exchanges per trip so the LDS reads pipeline
      for (int t0 = threadIdx.x; t0 < half; t0 += 4 * blockDim.x) {
        int ii[4], pp[4];
        unsigned long long a[4], b[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = t0 + u * blockDim.x;
          ii[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          pp[u] = mirror ? (ii[u] | (k - 1)) - (t & (j - 1)) : (ii[u] | j);
          live[u] = t < half && pp[u] < n;
          a[u] = live[u] ? s[ii[u]] : 0ull;
          b[u] = live[u] ? s[pp[u]] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (live[u] && a[u] < b[u]) {
            s[ii[u]] = b[u];
            s[pp[u]] = a[u];
          }
        }
      }
      __syncthreads();
    }
  }
}

struct GlobalKeys {
  const unsigned long long* p;
  __device__ __forceinline__ unsigned long long operator()(int t) const { return p[t]; }
};
struct ScoreKeys {  // keys built on the fly from a dense score array (merge stage)
  const float* s;
  __device__ __forceinline__ unsigned long long operator()(int t) const {
    return make_key(s[t], (unsigned int)t);
  }
};

// k-th largest (1-based) among keys[0..n) that are < upper.  Keys are unique.
template <class Loader>
__device__ unsigned long long radix_select_desc(Loader keys, int n, unsigned long long upper, int kth,
                                                int* hist /*256*/, unsigned long long* s_prefix, int* s_k,
                                                int* s_wtot /*4: per-wave bin totals*/) {
  if (threadIdx.x == 0) {
    *s_prefix = 0ull;
    *s_k = kth;
    s_k[3] = 0;   // (s_misc[3]) set when the selected bin is taken whole: the remaining digits do not matter
  }
  unsigned long long mask = 0ull;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int t = threadIdx.x; t < 256; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    const unsigned long long prefix = *s_prefix;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
      const unsigned long long key = keys(t);
      if (key < upper && (key & mask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
    }
    __syncthreads();
    {
      // the bin that holds the kk-th largest key: thread d owns bin d (the workgroup has 256 threads) and needs the number
      // of keys in the bins above it — a suffix sum inside the wave plus the totals of the waves above.  (One thread
      // walking the 256 bins cost ~5 us per pass, 8 passes per selection: a third of merge_kernel and a quarter of a
      // hard-NMS list at batch 8.)
      const int d = threadIdx.x, ln = d & 63, wv = d >> 6;
      const bool bin = d < 256;                    // (soft_nms_kernel runs 512 threads: the upper half only keeps the barriers)
      const int h = bin ? hist[d] : 0;
      const int kk = *s_k;
      int v = h;                                   // -> sum of the bins ln .. 63 of this wave
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_down(v, o, 64);
        if (ln + o < 64) v += u;
      }
      if (bin && ln == 0) s_wtot[wv] = v;
      __syncthreads();
      int above = v - h;
      for (int w = wv + 1; w < 4; ++w) above += s_wtot[w];
      // bin 0 takes what is left (as the serial walk did when the bins above hold fewer than kk keys)
      if (bin && above < kk && (d == 0 || kk <= above + h)) {
        *s_k = kk - above;
        *s_prefix = prefix | (((unsigned long long)d) << shift);
        // every key of this bin is among the kk largest: prefix (lower digits zero) is already the bound.  With distinct
        // scores that happens after the four or five score digits — the index digits of the 64-bit keys are never walked
        if (kk - above == h) s_k[3] = 1;
      }
    }
    mask |= 255ull << shift;
    __syncthreads();
    if (s_k[3]) break;
  }
  return *s_prefix;
}

// Load the `take` largest keys below `upper` into skeys (sorted descending) and return the
// smallest of them (the next chunk's exclusive upper bound; 0 when the list is exhausted).
template <class Loader>
__device__ unsigned long long next_chunk_sorted(Loader keys, int n, int remaining, int take,
                                                unsigned long long upper, unsigned long long* skeys,
                                                int* s_hist, unsigned long long* s_prefix, int* s_misc) {
  unsigned long long lower = 0ull;
  if (remaining > take) lower = radix_select_desc(keys, n, upper, take, s_hist, s_prefix, &s_misc[0], &s_misc[4]);
  if (threadIdx.x == 0) s_misc[1] = 0;
  for (int t = threadIdx.x; t < take; t += blockDim.x) skeys[t] = 0ull;
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const unsigned long long key = keys(t);
    if (key >= lower && key < upper) {
      const int slot = atomicAdd(&s_misc[1], 1);
      if (slot < take) skeys[slot] = key;
    }
  }
  __syncthreads();
  bitonic_sort_desc(skeys, take);
  return lower;
}

struct BoxSrc {
  const float4* base;
  long long img_stride, idx_stride, cls_stride;  // float4 units
};

__device__ __forceinline__ float4 fetch_clipped(const BoxSrc& bs, int b, int c, unsigned int idx) {
  float4 v = bs.base[(long long)b * bs.img_stride + (long long)idx * bs.idx_stride + (long long)c * bs.cls_stride];
  // tf.clip_by_value(boxes, 0, 1)  (postprocessing_ops.py:501)
  v.x = fminf(fmaxf(v.x, 0.0f), 1.0f);
  v.y = fminf(fmaxf(v.y, 0.0f), 1.0f);
  v.z = fminf(fmaxf(v.z, 0.0f), 1.0f);
  v.w = fminf(fmaxf(v.w, 0.0f), 1.0f);
  return v;
}

// TensorFlow NonMaxSuppression IOU: canonicalise corners, zero/negative area -> 0.
__device__ __forceinline__ float nms_iou(float4 a, float4 b) {
  const float ay0 = fminf(a.x, a.z), ax0 = fminf(a.y, a.w), ay1 = fmaxf(a.x, a.z), ax1 = fmaxf(a.y, a.w);
  const float by0 = fminf(b.x, b.z), bx0 = fminf(b.y, b.w), by1 = fmaxf(b.x, b.z), bx1 = fmaxf(b.y, b.w);
  const float area_a = (ay1 - ay0) * (ax1 - ax0);
  const float area_b = (by1 - by0) * (bx1 - bx0);
  if (area_a <= 0.0f || area_b <= 0.0f) return 0.0f;
  const float iy0 = fmaxf(ay0, by0), ix0 = fmaxf(ax0, bx0);
  const float iy1 = fminf(ay1, by1), ix1 = fminf(ax1, bx1);
  const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
  return inter / (area_a + area_b - inter);
}

__device__ __forceinline__ float4 shfl_box(float4 v, int src) {
  float4 r;
  r.x = __shfl(v.x, src, 64);
  r.y = __shfl(v.y, src, 64);
  r.z = __shfl(v.z, src, 64);
  r.w = __shfl(v.w, src, 64);
  return r;
}

// max over the 64 lanes of a wave, returned to every lane: six DPP steps (quad swaps, row rotations, row broadcasts — the
// gfx9 wave-reduction sequence) and one v_readlane; a __shfl_xor tree is six ds_bpermute round trips through the LDS crossbar
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v) {
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ float wave_max_f32(float v) {
  v = dpp_max_step<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v = dpp_max_step<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
  v = dpp_max_step<0x124, 0xf>(v);   // row_ror:4
  v = dpp_max_step<0x128, 0xf>(v);   // row_ror:8   -> every lane holds its row's max
  v = dpp_max_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v = dpp_max_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's max
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// One workgroup per (image, class).  Dynamic LDS: keys[chunk_cap] | sel boxes | sel scores | sel indices (sel_cap each)
// | hist | soft-NMS state.
struct NmsParams {
  int B, K, max_det, top_k;
  float iou_thr, score_thr, soft_scale;  // soft_scale = -0.5/(sigma/2) when soft, else 0
  int soft;
  long long cap;
  int soft_off;         // soft: byte offset of the soft-NMS state in LDS (behind everything else)
  int chunk_cap;        // keys the LDS sort buffer holds: soft NMS, the whole list (one chunk, <= RN_SORT_CAP); RN_HARD_CHUNK else
  int sel_cap;          // slots of the selected-box arrays (max_det rounded up to 4)
  int part_off, first_chunk;   // hard: byte offset of the four waves' partial results (32 + 512 B); keys of the first sorted chunk
};

__global__ void __launch_bounds__(RN_PP_THREADS)
nms_per_class_kernel(NmsParams p, const int* __restrict__ counts, const unsigned long long* __restrict__ keys_g,
                     BoxSrc bs, float* __restrict__ sel_scores, float4* __restrict__ sel_boxes,
                     int* __restrict__ sel_idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* skeys = (unsigned long long*)smem;                       // chunk_cap * 8
  float4* s_selbox = (float4*)(smem + (size_t)p.chunk_cap * 8);                // sel_cap * 16
  float* s_selscore = (float*)((char*)s_selbox + (size_t)p.sel_cap * 16);      // sel_cap * 4
  int* s_selidx = (int*)((char*)s_selscore + (size_t)p.sel_cap * 4);           // sel_cap * 4
  int* s_hist = (int*)((char*)s_selidx + (size_t)p.sel_cap * 4);               // 256 * 4
  unsigned long long* s_prefix = (unsigned long long*)((char*)s_hist + 1024);  // 8
  int* s_misc = (int*)((char*)s_prefix + 8);                                   // [0]=k scratch [1]=fill [2]=nsel [4..7]=wave totals (radix select)
  // soft NMS state (sized by nms_lds_bytes below): two cached blocks of candidate boxes | per block (best score, its
  // position) | one byte per candidate.  A candidate's CURRENT score lives in the score half of its sorted key.
  float4* s_cbox = (float4*)(smem + p.soft_off);                               // 2 * 64 * 16
  int2* s_blk = (int2*)((char*)s_cbox + 2048);                                 // RN_SORT_CAP / 64 * 8: (best score, its position)
  unsigned char* s_sbi = (unsigned char*)((char*)s_blk + RN_SORT_CAP / 64 * 8 + 16);   // candidates (<= max_det - 1 <= 255 each)
  float* s_cur2 = (float*)skeys;                                               // score of candidate i: s_cur2[2 * i + 1]

  const int list = blockIdx.x;  // b*K + c
  const int b = list / p.K, c = list - b * p.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long n_ll = counts[list];
  if (n_ll > p.cap) n_ll = p.cap;
  const int n = (int)n_ll;
  const unsigned long long* keys = keys_g + (long long)list * p.cap;
  int limit = n;
  if (p.top_k > 0 && limit > p.top_k) limit = p.top_k;
  if (p.soft && limit > p.chunk_cap) limit = p.chunk_cap;  // host rejects configs that could reach this

  if (threadIdx.x == 0) s_misc[2] = 0;
  __syncthreads();

  unsigned long long upper = ~0ull;
  int processed = 0;
  const GlobalKeys kload{keys};
  while (processed < limit) {
    if (s_misc[2] >= p.max_det) break;
    // hard NMS usually fills max_det from the first few hundred candidates: sort a small first
    // chunk (NmsParams.first_chunk), fall back to full LDS-sized chunks only when suppression eats through it
    int cap_chunk = p.chunk_cap;
    if (!p.soft && processed == 0) cap_chunk = p.first_chunk;
    const int take = (limit - processed) < cap_chunk ? (limit - processed) : cap_chunk;
    const unsigned long long lower =
        next_chunk_sorted(kload, n, n - processed, take, upper, skeys, s_hist, s_prefix, s_misc);
    const int m = take;

    if (!p.soft) {
      // ---- greedy hard NMS, 64 candidates per step, all four waves ---------------------------------
      // Every wave holds the group (one candidate per lane).  The two quadratic parts are split four ways — wave w tests
      // its candidates against the selected boxes w, w + 4, ... and against the group's lanes 16w .. 16w + 15 — and meet in
      // LDS; wave 0 then runs the sequential selection.  (Measured: batch-8 inference 3.76 -> 3.74 ms.  Tried beside it and
      // dropped: wave-aggregated histogram updates in the radix select — the high-byte passes put every key of a list into
      // one bin — 3.76 -> 3.90 ms: the ballots in the loop keep the key loads from overlapping.)
      unsigned long long* s_av = (unsigned long long*)(smem + p.part_off);       // [4] alive lanes per wave
      unsigned short* s_pm = (unsigned short*)(smem + p.part_off + 32);         // [4][64] 16 mask bits per wave and lane
      float4* s_gbox = (float4*)s_hist;                                         // [64] the group's boxes (histogram idle here)
      int nsel = s_misc[2];
      for (int base = 0; base < m && nsel < p.max_det; base += 64) {
        const int i = base + lane;
        const bool valid = i < m;
        const unsigned long long key = valid ? skeys[i] : 0ull;
        const float score = key_score(key);
        const unsigned int idx = key_index(key);
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) box = fetch_clipped(bs, b, c, idx);
        bool alive = valid;
        for (int j = wave; j < nsel; j += 4) {
          const float4 sb = s_selbox[j];
          if (alive && nms_iou(box, sb) > p.iou_thr) alive = false;
        }
        // earlier lanes of the group that suppress me: the wave's 16 lanes of the group go through its own slice of LDS
        // (independent broadcast reads pipeline; a shuffle of lane j's box is four ds_bpermute round trips per step)
        if ((lane >> 4) == wave) s_gbox[lane] = box;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        unsigned int part = 0u;
        for (int q = 0; q < 16; ++q) {
          const int j = wave * 16 + q;
          const float4 ob = s_gbox[j];
          if (j < lane && nms_iou(box, ob) > p.iou_thr) part |= 1u << q;
        }
        const unsigned long long av = __ballot(alive);
        if (lane == 0) s_av[wave] = av;
        s_pm[wave * 64 + lane] = (unsigned short)part;
        __syncthreads();
        if (wave == 0) {
          unsigned long long alive_mask = s_av[0] & s_av[1] & s_av[2] & s_av[3];
          const unsigned long long mask = (unsigned long long)s_pm[lane] | ((unsigned long long)s_pm[64 + lane] << 16) |
                                          ((unsigned long long)s_pm[128 + lane] << 32) | ((unsigned long long)s_pm[192 + lane] << 48);
          for (int j = 0; j < 64; ++j) {
            if (!((alive_mask >> j) & 1ull)) continue;
            if (nsel >= p.max_det) break;
            if (lane == j) {
              s_selbox[nsel] = box;
              s_selscore[nsel] = score;
              s_selidx[nsel] = (int)idx;
            }
            ++nsel;
            const unsigned long long kill = __ballot((mask >> j) & 1ull);
            alive_mask &= ~kill;
          }
          if (lane == 0) s_misc[2] = nsel;
        }
        __syncthreads();
        nsel = s_misc[2];
      }
    } else if (wave == 0) {
      int nsel = s_misc[2];
      {
        // ---- soft NMS: exact emulation of NonMaxSuppressionV5's priority queue ------------
        // The queue is the sorted chunk itself: candidate i keeps its position, its CURRENT score replaces the score half
        // of its key (-1: popped for good, or fell to the threshold).  Per block j of 64 candidates: s_blk[j] = (the
        // block's best score, the lowest position inside the block holding it).  A pop is the argmax over the
        // block maxima (lowest block on ties) — the queue's (score desc, position asc) order — in a handful of LDS reads and
        // two DPP wave reductions.  The boxes of a block's 64 candidates are fetched together, one per lane, when the block
        // first leads (two blocks cached: pops walk the sorted order, re-inserted candidates sit in the blocks behind).
        // Round 4: until then every pop scanned all m candidates (~60 dependent LDS reads), multiplied the weights
        // through up to 100 dependent ds_bpermute's and fetched its box from memory; on clustered detections — a pop per
        // candidate, most of them decayed to death by their selected neighbours — EfficientNet-B3's batch-8 serving
        // step spent 140 of its 147 ms in this loop (now ~11: 2 000 - 8 000 pops per class at ~2 200 cycles each, a chain
        // of dependent LDS reads and VALU operations on one wave).  The LDS footprint dropped from 122 KB to 73 KB (state
        // in the sort buffer's unused tail): two lists per compute unit.
        // Tried and dropped: pops in batches (one candidate of the leading block per lane, re-scored in parallel, a scalar
        // walk in queue order up to the next selection — pops between two selections are independent).  Bit-identical, 17
        // pops per batch on the clustered lists, and slower (27 - 50 ms): re-inserted candidates make the queue's top jump
        // between many blocks, each visit needing that block's boxes and overlap masks for one or two pops.
        const int nblk = (m + 63) >> 6;
        for (int j = 0; j < nblk; ++j) {
          const int i = j * 64 + lane;
          float sc = -1.0f;
          if (i < m) {
            sc = key_score(skeys[i]);
            s_sbi[i] = 0;
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);   // the key is read before its score half is rewritten
          if (i < m) s_cur2[2 * i + 1] = sc;
          const float mx = wave_max_f32(sc);
          const int pos = (int)__builtin_ctzll(__ballot(sc == mx));
          if (lane == 0) s_blk[j] = make_int2(__float_as_int(mx), pos);
        }
        int tag0 = -1, tag1 = -1;   // the blocks whose boxes sit in the two cache slots (wave-uniform)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        while (nsel < p.max_det) {
          // lane l looks at blocks l and l + 64 (m <= RN_SORT_CAP = 128 blocks); the lowest block holding the maximum
          const int2 e0 = lane < nblk ? s_blk[lane] : make_int2(__float_as_int(-1.0f), 0);
          const int2 e1 = lane + 64 < nblk ? s_blk[lane + 64] : make_int2(__float_as_int(-1.0f), 0);
          const float b0 = __int_as_float(e0.x), b1 = __int_as_float(e1.x);
          const float best = wave_max_f32(fmaxf(b0, b1));
          if (!(best > p.score_thr)) break;  // queue empty
          const unsigned long long h0 = __ballot(b0 == best), h1 = __ballot(b1 == best);
          const int bl = __builtin_amdgcn_readfirstlane(h0 ? (int)__builtin_ctzll(h0) : (int)__builtin_ctzll(h1));
          const int blk = h0 ? bl : 64 + bl;
          const int pos = h0 ? __builtin_amdgcn_readlane(e0.y, bl) : __builtin_amdgcn_readlane(e1.y, bl);
          const int i0 = blk * 64 + lane;
          const int e = blk & 1;
          if ((e ? tag1 : tag0) != blk) {   // (uniform) this block's boxes: 64 fetches in flight instead of one per pop
            float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i0 < m) bx = fetch_clipped(bs, b, c, key_index(skeys[i0]));
            s_cbox[e * 64 + lane] = bx;
            if (e) tag1 = blk; else tag0 = blk;
            __builtin_amdgcn_s_waitcnt(0xc07f);
          }
          const int besti = blk * 64 + pos;
          const float4 box = s_cbox[e * 64 + pos];
          const float mine = i0 < m ? s_cur2[2 * i0 + 1] : -1.0f;
          const int begin = s_sbi[besti];
          float score = best;
          // newest -> oldest; weights computed 64 at a time, multiplied in order
          for (int hi = nsel - 1; hi >= begin && score > p.score_thr; hi -= 64) {
            const int j = hi - lane;
            float w = 1.0f;
            if (j >= begin) {
              const float sim = nms_iou(box, s_selbox[j]);
              w = rn_expf(p.soft_scale * sim * sim);
            }
            // the weights multiply the score in lane order (newest selected box first); a weight of exactly 1 (no overlap:
            // the common case) leaves the score as it is, so only the other lanes are visited — each step one v_readlane
            unsigned long long nz = __ballot(w != 1.0f);
            while (nz) {
              const int q = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(nz));
              nz &= nz - 1;
              score = score * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), q));
              if (score <= p.score_thr) break;
            }
          }
          // selected (score unchanged), re-inserted with its lower score, or dropped
          const float back = (score == best || !(score > p.score_thr)) ? -1.0f : score;
          const float now = lane == pos ? back : mine;       // the block's scores after this pop
          const float mx = wave_max_f32(now);
          const int npos = (int)__builtin_ctzll(__ballot(now == mx));
          if (lane == 0) {
            s_sbi[besti] = (unsigned char)nsel;
            if (score == best) {
              s_selbox[nsel] = box;
              s_selscore[nsel] = score;
              s_selidx[nsel] = (int)key_index(skeys[besti]);
            }
            s_cur2[2 * besti + 1] = back;
            s_blk[blk] = make_int2(__float_as_int(mx), npos);
          }
          if (score == best) ++nsel;
          __builtin_amdgcn_s_waitcnt(0xc07f);
        }
      }
      if (lane == 0) s_misc[2] = nsel;
    }
    __syncthreads();
    upper = lower;
    processed += take;
    if (lower == 0ull) break;
  }
  __syncthreads();
  // write padded per-class results (NonMaxSuppressionV5 pads scores with 0)
  const int nsel = s_misc[2];
  for (int t = threadIdx.x; t < p.max_det; t += blockDim.x) {
    const bool ok = t < nsel;
    sel_scores[(long long)list * p.max_det + t] = ok ? s_selscore[t] : 0.0f;
    sel_boxes[(long long)list * p.max_det + t] = ok ? s_selbox[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    sel_idx[(long long)list * p.max_det + t] = ok ? s_selidx[t] : 0;
  }
}

// ---- soft NMS, one selection per step ------------------------------------------------------------------------------
// NonMaxSuppressionV5's soft path is a lazy queue: a popped candidate is re-scored against the boxes selected since it was
// last looked at (newest first, one float multiply per box); unchanged -> selected, lower -> pushed back.  On clustered
// detections that is ~80 pops per selection, each a chain of dependent LDS reads and reductions on one wave (the loop in
// nms_per_class_kernel above: 8 000 pops x ~2 200 cycles = 8 ms for one list of the configs[4] serving bench, with the
// other three waves of the workgroup idle).  The same results — the same candidates in the same order with the same score
// bits — come out of a form with ONE step per selected box and every thread busy:
//   * a thread owns NC candidates (positions tid, tid + THREADS, ...: box, queue score q, and the weights != 1 of the boxes
//     selected since the candidate was last re-scored — newest first, up to 4 — all in registers);
//   * what the queue would hand back for a candidate now is r = q * w_newest * ... * w_oldest (the weights that are
//     exactly 1 — no overlap, the common case — do not change a float product, so leaving them out is exact);
//   * the next selected box is the candidate with the largest r (lowest position on ties: the queue's order), R = its r.
//     Proof sketch: the queue pops in descending q; everything popped before the selection has q > R (or q = R and a lower
//     position), is re-scored to its r <= R and pushed back; the first pop that comes back unchanged is that maximum;
//   * exactly the candidates the queue popped on the way — q > R, or q = R at a lower position — take q := r and forget
//     their weights; r <= score_threshold drops a candidate (the queue drops it when it pops it; it is never selected in
//     between, so dropping it early changes nothing);
//   * every live candidate then gets the weight of the new box; a fifth weight != 1 turns the candidate into the general
//     case: its r is recomputed over the selected boxes [begin, nsel) as the queue does — only when it is popped (q > R),
//     which is rare (tools/probes/soft_nms_sim.py: 1 - 500 of ~8 000 re-scorings per list).
// Verified against the queue (oracle rn_o_nms_v5) bit for bit by the soft-NMS tests of tests/test_gpu_postprocess.py and
// tests/test_soft_nms_independent.py.  Reference: postprocessing_ops.py:443-451 (tf.image.non_max_suppression_with_scores).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_step(int v) {
  const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = dpp_min_step<0xb1, 0xf>(v);
  v = dpp_min_step<0x4e, 0xf>(v);
  v = dpp_min_step<0x124, 0xf>(v);
  v = dpp_min_step<0x128, 0xf>(v);
  v = dpp_min_step<0x142, 0xa>(v);
  v = dpp_min_step<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
#define RN_SOFT_PEND 4
#define RN_SOFT_OVER 5      // more than RN_SOFT_PEND weights != 1 since the last re-scoring
#define RN_SOFT_EVAL 6      // general case, re-scored in this step (the value sits in pend[0])
// a candidate's box with its corners in order (nms_iou's first step).
// Boxes reach this kernel through fetch_clipped: finite, in [0, 1] — min / max are written as compare + select (fminf / fmaxf
// on values the compiler cannot prove canonical cost a v_max x, x each; the two forms differ for NaN and the sign of zero only,
// and a zero of either sign leaves the weight at exp(0) = 1)
struct SoftBox { float y0, x0, y1, x1; };
__device__ __forceinline__ float sel_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float sel_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ SoftBox soft_box(const float4 a) {
  SoftBox s;
  s.y0 = sel_min(a.x, a.z); s.x0 = sel_min(a.y, a.w); s.y1 = sel_max(a.x, a.z); s.x1 = sel_max(a.y, a.w);
  return s;
}
// rn_expf for x <= 0 (not NaN): the same operations without the upper range tests
__device__ __forceinline__ float soft_expf_nonpos(float x) {
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float q = 1.9875691500e-4f;
  q = fmaf(q, r, 1.3981999507e-3f);
  q = fmaf(q, r, 8.3334519073e-3f);
  q = fmaf(q, r, 4.1665795894e-2f);
  q = fmaf(q, r, 1.6666665459e-1f);
  q = fmaf(q, r, 5.0000001201e-1f);
  const float r2 = r * r;
  q = fmaf(q, r2, r);
  q = q + 1.0f;
  const float e = ldexpf(q, (int)n);
  return x < -103.972084f ? 0.0f : e;
}
// exp(scale * iou^2), scale <= 0, with the arithmetic of nms_iou / rn_expf operation for operation
__device__ __forceinline__ float soft_weight(const SoftBox& a, const SoftBox& b, float scale) {
  const float iy0 = sel_max(a.y0, b.y0), ix0 = sel_max(a.x0, b.x0);
  const float iy1 = sel_min(a.y1, b.y1), ix1 = sel_min(a.x1, b.x1);
  const float inter = sel_max(iy1 - iy0, 0.0f) * sel_max(ix1 - ix0, 0.0f);
  const float area_a = (a.y1 - a.y0) * (a.x1 - a.x0), area_b = (b.y1 - b.y0) * (b.x1 - b.x0);
  float sim = inter / (area_a + area_b - inter);
  sim = (area_a <= 0.0f || area_b <= 0.0f) ? 0.0f : sim;
  return soft_expf_nonpos(scale * sim * sim);
}
template <int NC, int THREADS>
__global__ void __launch_bounds__(THREADS, 4)
soft_nms_kernel(NmsParams p, const int* __restrict__ counts, const unsigned long long* __restrict__ keys_g,
                BoxSrc bs, float* __restrict__ sel_scores, float4* __restrict__ sel_boxes, int* __restrict__ sel_idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* skeys = (unsigned long long*)smem;                       // chunk_cap * 8
  float4* s_selbox = (float4*)(smem + (size_t)p.chunk_cap * 8);                // sel_cap * 16
  float* s_selscore = (float*)((char*)s_selbox + (size_t)p.sel_cap * 16);      // sel_cap * 4
  int* s_selidx = (int*)((char*)s_selscore + (size_t)p.sel_cap * 4);           // sel_cap * 4
  int* s_hist = (int*)((char*)s_selidx + (size_t)p.sel_cap * 4);               // 256 * 4
  unsigned long long* s_prefix = (unsigned long long*)((char*)s_hist + 1024);  // 8
  int* s_misc = (int*)((char*)s_prefix + 8);                                   // 32
  float* s_red = (float*)(smem + p.soft_off);                                  // [2][THREADS / 64][4]: per wave (best r, its position, max q of the general case, -)

  const int list = blockIdx.x;
  const int b = list / p.K, cls = list - b * p.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long n_ll = counts[list];
  if (n_ll > p.cap) n_ll = p.cap;
  const int n = (int)n_ll;
  int m = n;
  if (p.top_k > 0 && m > p.top_k) m = p.top_k;
  if (m > p.chunk_cap) m = p.chunk_cap;   // host rejects configs that could reach this
  int nsel = 0;
  if (m > 0) {
    next_chunk_sorted(GlobalKeys{keys_g + (long long)list * p.cap}, n, n, m, ~0ull, skeys, s_hist, s_prefix, s_misc);
    const int ncu = (m + THREADS - 1) / THREADS;   // candidates per thread in use (uniform)
    const float thr = p.score_thr;
    SoftBox bx[NC];
    float q[NC], pend[NC][RN_SOFT_PEND];   // q: the candidate's score in the queue (-1: gone); pend: weights != 1, newest first
    int meta[NC];   // bits 0..2: weights held (0..4) | RN_SOFT_OVER | RN_SOFT_EVAL; bit 3 / 4: the box came with y0 > y1 / x0 > x1; bits 8..: begin
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = c * THREADS + (int)threadIdx.x;
      q[c] = -1.0f;
      meta[c] = 0;
      bx[c] = soft_box(make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
      for (int k = 0; k < RN_SOFT_PEND; ++k) pend[c][k] = 1.0f;
      if (c < ncu && i < m) {
        const unsigned long long key = skeys[i];
        const float sc = key_score(key);
        if (sc > thr) {
          q[c] = sc;
          const float4 o = fetch_clipped(bs, b, cls, key_index(key));
          bx[c] = soft_box(o);
          meta[c] = (o.x > o.z ? 8 : 0) | (o.y > o.w ? 16 : 0);
        }
      }
    }
    // the first selection is the head of the sorted list
    float R = key_score(skeys[0]);
    int mpos = 0, par = 0;
    while (R > thr) {
      // the selected candidate's owner records it (its box as it came)
      if ((mpos & (THREADS - 1)) == (int)threadIdx.x) {
        const int cm = mpos / THREADS;
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (c == cm) {
            const bool sy = meta[c] & 8, sx = meta[c] & 16;
            s_selbox[nsel] = make_float4(sy ? bx[c].y1 : bx[c].y0, sx ? bx[c].x1 : bx[c].x0, sy ? bx[c].y0 : bx[c].y1,
                                         sx ? bx[c].x0 : bx[c].x1);
          }
        s_selscore[nsel] = R;
        s_selidx[nsel] = (int)key_index(skeys[mpos]);
      }
      __syncthreads();
      const SoftBox nb = soft_box(s_selbox[nsel]);
      const bool last = nsel + 1 >= p.max_det;
      // one pass over this thread's candidates: (a) the ones the queue popped on its way to this selection take their new
      // scores, (b) the weight of the new box, (c) what the queue would return now -> this thread's best
      float br = -1.0f, oq = -1.0f;
      int bp = 0x7fffffff;
      if (!last) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          __builtin_amdgcn_sched_barrier(0);   // one candidate's temporaries at a time: the state fills the register file
          if (c < ncu) {
            const int i = c * THREADS + (int)threadIdx.x;
            int mt = meta[c];
            float qc = q[c];
            {
              const int st = mt & 7;
              float r = qc;
#pragma unroll
              for (int k = 0; k < RN_SOFT_PEND; ++k) r = r * pend[c][k];
              r = st == RN_SOFT_EVAL ? pend[c][0] : r;
              const bool pop = qc > thr && (qc > R || (qc == R && i < mpos));
              qc = i == mpos ? -1.0f : (pop ? r : qc);
#pragma unroll
              for (int k = 0; k < RN_SOFT_PEND; ++k) pend[c][k] = pop ? 1.0f : pend[c][k];
              // popped: weights forgotten, begin = nsel; looked at but not popped: back to the general case
              mt = pop ? (mt & 0x18) | (nsel << 8) : (st == RN_SOFT_EVAL ? (mt & ~7) | RN_SOFT_OVER : mt);
            }
            const bool live = qc > thr;
            {
              const float w = soft_weight(bx[c], nb, p.soft_scale);
              const int st = mt & 7;
              const bool add = w != 1.0f && live && st <= RN_SOFT_PEND;   // a fifth weight: the general case from here on
              const bool keep = add && st < RN_SOFT_PEND;
#pragma unroll
              for (int k = RN_SOFT_PEND - 1; k > 0; --k) pend[c][k] = keep ? pend[c][k - 1] : pend[c][k];
              pend[c][0] = keep ? w : pend[c][0];
              mt += add ? 1 : 0;
            }
            {
              float r = qc;
#pragma unroll
              for (int k = 0; k < RN_SOFT_PEND; ++k) r = r * pend[c][k];   // (once <= threshold it stays there: weights <= 1)
              const bool general = (mt & 7) > RN_SOFT_PEND;
              const bool ok = live && !general && r > thr;
              qc = (live && !general && !(r > thr)) ? -1.0f : qc;          // the queue drops it when it pops it
              const bool take = ok && r > br;
              br = take ? r : br;
              bp = take ? i : bp;
              oq = sel_max(oq, (live && general) ? qc : -1.0f);
            }
            q[c] = qc;
            meta[c] = mt;
          }
        }
      }
      ++nsel;
      if (last) break;
      float* red = s_red + par * (THREADS / 16);
      par ^= 1;
      {
        const float wm = wave_max_f32(br);
        const int wp = wave_min_i32(br == wm ? bp : 0x7fffffff);
        const float wo = wave_max_f32(oq);
        if (lane == 0) {
          red[wave * 4 + 0] = wm;
          red[wave * 4 + 1] = __int_as_float(wp);
          red[wave * 4 + 2] = wo;
        }
      }
      __syncthreads();
      float OQ = -1.0f;
      R = -1.0f;
      mpos = 0x7fffffff;
#pragma unroll
      for (int w = 0; w < THREADS / 64; ++w) {
        const float wm = red[w * 4 + 0];
        const int wp = __float_as_int(red[w * 4 + 1]);
        OQ = sel_max(OQ, red[w * 4 + 2]);
        if (wm > R || (wm == R && wp < mpos)) { R = wm; mpos = wp; }
      }
      if (OQ >= R && OQ > thr) {   // (uniform) general-case candidates the queue would pop before the selection
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          __builtin_amdgcn_sched_barrier(0);
          const int i = c * THREADS + (int)threadIdx.x;
          if (c < ncu && q[c] > thr && (meta[c] & 7) == RN_SOFT_OVER && (q[c] > R || (q[c] == R && i < mpos))) {
            float sc = q[c];
            for (int j = nsel - 1; j >= (meta[c] >> 8) && sc > thr; --j)
              sc = sc * soft_weight(bx[c], soft_box(s_selbox[j]), p.soft_scale);
            if (!(sc > thr)) q[c] = -1.0f;
            else {
              pend[c][0] = sc;
              meta[c] = (meta[c] & ~7) | RN_SOFT_EVAL;
              if (sc > br || (sc == br && i < bp)) { br = sc; bp = i; }
            }
          }
        }
        red = s_red + par * (THREADS / 16);
        par ^= 1;
        const float wm = wave_max_f32(br);
        const int wp = wave_min_i32(br == wm ? bp : 0x7fffffff);
        if (lane == 0) {
          red[wave * 4 + 0] = wm;
          red[wave * 4 + 1] = __int_as_float(wp);
        }
        __syncthreads();
        R = -1.0f;
        mpos = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) {
          const float wm2 = red[w * 4 + 0];
          const int wp2 = __float_as_int(red[w * 4 + 1]);
          if (wm2 > R || (wm2 == R && wp2 < mpos)) { R = wm2; mpos = wp2; }
        }
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < p.max_det; t += blockDim.x) {
    const bool ok = t < nsel;
    sel_scores[(long long)list * p.max_det + t] = ok ? s_selscore[t] : 0.0f;
    sel_boxes[(long long)list * p.max_det + t] = ok ? s_selbox[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    sel_idx[(long long)list * p.max_det + t] = ok ? s_selidx[t] : 0;
  }
}

// ---- merge: top max_det over K*max_det padded class results (postprocessing_ops.py:471-490)
__global__ void __launch_bounds__(RN_PP_THREADS)
merge_kernel(int K, int max_det, const float* __restrict__ sel_scores, const float4* __restrict__ sel_boxes,
             const int* __restrict__ sel_idx, float4* __restrict__ det_boxes, float* __restrict__ det_scores,
             int* __restrict__ det_classes, int* __restrict__ det_index, int* __restrict__ valid_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* skeys = (unsigned long long*)smem;               // RN_MAX_DET * 8
  int* s_hist = (int*)(smem + (size_t)RN_MAX_DET * 8);                 // 1024
  unsigned long long* s_prefix = (unsigned long long*)((char*)s_hist + 1024);
  int* s_misc = (int*)((char*)s_prefix + 8);                           // [0..1] scratch, [2] valid
  const int b = blockIdx.x;
  const int total = K * max_det;
  const ScoreKeys kload{sel_scores + (long long)b * total};
  if (threadIdx.x == 0) s_misc[2] = 0;
  next_chunk_sorted(kload, total, total, max_det, ~0ull, skeys, s_hist, s_prefix, s_misc);
  int cnt = 0;
  for (int t = threadIdx.x; t < max_det; t += blockDim.x) {
    const float s = key_score(skeys[t]);
    if (s > 0.0f) ++cnt;
  }
  if (cnt) atomicAdd(&s_misc[2], cnt);
  __syncthreads();
  const int valid = s_misc[2];
  for (int t = threadIdx.x; t < max_det; t += blockDim.x) {
    const unsigned long long key = skeys[t];
    const float s = key_score(key);
    const unsigned int flat = key_index(key);
    const bool ok = t < valid;
    det_scores[(long long)b * max_det + t] = ok ? s : -1.0f;
    det_classes[(long long)b * max_det + t] = ok ? (int)(flat / (unsigned int)max_det) : -1;
    det_boxes[(long long)b * max_det + t] =
        ok ? sel_boxes[(long long)b * total + flat] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (det_index) det_index[(long long)b * max_det + t] = ok ? sel_idx[(long long)b * total + flat] : -1;
  }
  if (threadIdx.x == 0) valid_out[b] = valid;
}

// ---- top-k emit: sorted chunks written straight to [B,k,K] ---------------------------------
__global__ void __launch_bounds__(RN_PP_THREADS)
topk_emit_kernel(int K, int k_out, const int* __restrict__ counts, const unsigned long long* __restrict__ keys_g,
                 long long cap, float* __restrict__ out_scores, int* __restrict__ out_idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* skeys = (unsigned long long*)smem;
  int* s_hist = (int*)(smem + (size_t)RN_SORT_CAP * 8);
  unsigned long long* s_prefix = (unsigned long long*)((char*)s_hist + 1024);
  int* s_misc = (int*)((char*)s_prefix + 8);
  const int list = blockIdx.x;
  const int b = list / K, c = list - b * K;
  long long n_ll = counts[list];
  if (n_ll > cap) n_ll = cap;
  const int n = (int)n_ll;
  const unsigned long long* keys = keys_g + (long long)list * cap;
  const int limit = n < k_out ? n : k_out;
  unsigned long long upper = ~0ull;
  int processed = 0;
  const GlobalKeys kload{keys};
  while (processed < limit) {
    const int take = (limit - processed) < RN_SORT_CAP ? (limit - processed) : RN_SORT_CAP;
    const unsigned long long lower =
        next_chunk_sorted(kload, n, n - processed, take, upper, skeys, s_hist, s_prefix, s_misc);
    for (int t = threadIdx.x; t < take; t += blockDim.x) {
      const unsigned long long key = skeys[t];
      const long long o = ((long long)b * k_out + processed + t) * K + c;
      out_scores[o] = key_score(key);
      out_idx[o] = (int)key_index(key);
    }
    __syncthreads();
    upper = lower;
    processed += take;
    if (lower == 0ull) break;
  }
}

// ---- max / argmax over classes per row (global NMS modes, postprocessing_ops.py:248-261) ---------
__global__ void __launch_bounds__(RN_PP_THREADS)
rowmax_kernel(const float* __restrict__ scores, long long rows, int K, float* __restrict__ mx, int* __restrict__ arg) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < rows;
       r += (long long)gridDim.x * blockDim.x) {
    const float* p = scores + r * K;
    float best = p[0];
    int bi = 0;
    for (int k = 1; k < K; ++k)
      if (p[k] > best) {  // tf.argmax / reduce_max: first maximum
        best = p[k];
        bi = k;
      }
    mx[r] = best;
    arg[r] = bi;
  }
}
extern "C" int rn_rowmax_argmax(const float* scores, int64_t rows, int K, float* max_out, int32_t* argmax_out,
                                void* stream) {
  RN_CHECK_ARG(scores && max_out && argmax_out && rows > 0 && K > 0, "rn_rowmax_argmax: bad argument");
  hipLaunchKernelGGL(rowmax_kernel, dim3(pp_blocks(rows)), dim3(RN_PP_THREADS), 0, (hipStream_t)stream, scores,
                     (long long)rows, K, max_out, argmax_out);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ------------------------------------------------------------------------------------------
// LDS of nms_per_class_kernel.  Soft NMS sorts its whole list (at most `max_cands` <= RN_SORT_CAP keys) as one chunk and
// keeps its state (2 KB box cache, 1 KB block maxima / leaders, one byte per candidate) behind the other arrays:
// *soft_off = its byte offset.  With pre_nms_top_k = 5 000 and max_det = 100 (the reference's soft-NMS configurations) that
// is 52 KB — three lists per compute unit, the 640 lists of a batch of 8 resident at once; the sort buffer of 8 192 keys
// (the bitonic network's power of two, before bitonic_sort_desc learnt other sizes) made it 73 KB: two per unit, two rounds.
// Hard NMS takes its candidates in sorted chunks (1 024 first — it usually fills max_det from a few hundred — then
// RN_HARD_CHUNK at a time): 4 096 keys = 39 KB of LDS = four lists per compute unit, where RN_SORT_CAP keys (73 KB) allowed two:
// the 640 lists of a batch of 8 run in one round instead of two.
#define RN_HARD_CHUNK 4096
#define RN_SOFT_NC 10        // soft_nms_kernel: candidates per thread,
#define RN_SOFT_THREADS 512  // threads per list: lists of up to 5 120 candidates, ~100 registers of state per thread
static size_t nms_lds_bytes(int soft /* 0 hard, 1 nms_per_class_kernel's queue loop, 2 soft_nms_kernel */, long long max_cands,
                            int max_det, int* soft_off, int* chunk_cap, int* sel_cap, int* part_off) {
  const size_t cands = rn_align_up((size_t)(max_cands < RN_SORT_CAP ? max_cands : RN_SORT_CAP), 64);
  *chunk_cap = soft ? (int)cands : RN_HARD_CHUNK;
  *sel_cap = (int)rn_align_up((size_t)max_det, 4);
  const size_t base = rn_align_up((size_t)*chunk_cap * 8 + (size_t)*sel_cap * 24 + 1024 + 8 + 32, 16);
  *soft_off = 0;
  *part_off = (int)base;
  if (!soft) return base + 32 + 512;   // (4 x <= 40 520 B: four lists fit the 160 KB of a CU)
  const size_t state = soft == 2 ? 2 * (RN_SOFT_THREADS / 64) * 16 : rn_align_up(2048 + (size_t)RN_SORT_CAP / 64 * 8 + 16 + cands, 16);
  *soft_off = (int)base;
  return base + state;
}

struct DetectWs {
  int* counts;
  unsigned long long* keys;
  float* sel_scores;
  float4* sel_boxes;
  int* sel_idx;
  size_t total;
};

static DetectWs detect_ws_layout(void* base, int B, long long cap, int K, int max_det) {
  DetectWs w;
  size_t off = 0;
  char* p = (char*)base;
  w.counts = (int*)(p + off);
  off += rn_align_up((size_t)B * K * 4, 256);
  w.keys = (unsigned long long*)(p + off);
  off += rn_align_up((size_t)B * K * cap * 8, 256);
  w.sel_scores = (float*)(p + off);
  off += rn_align_up((size_t)B * K * max_det * 4, 256);
  w.sel_boxes = (float4*)(p + off);
  off += rn_align_up((size_t)B * K * max_det * 16, 256);
  w.sel_idx = (int*)(p + off);
  off += rn_align_up((size_t)B * K * max_det * 4, 256);
  w.total = off;
  return w;
}

extern "C" size_t rn_detect_workspace_bytes(int B, int64_t A, int K, int max_det) {
  if (B <= 0 || A <= 0 || K <= 0 || max_det <= 0) return 0;
  return detect_ws_layout(nullptr, B, A, K, max_det).total;
}
extern "C" size_t rn_nms_workspace_bytes(int B, int n, int K, int max_det) {
  return rn_detect_workspace_bytes(B, n, K, max_det);
}
extern "C" size_t rn_topk_workspace_bytes(int B, int64_t A, int K) {
  if (B <= 0 || A <= 0 || K <= 0) return 0;
  return rn_align_up((size_t)B * K * 4, 256) + rn_align_up((size_t)B * K * A * 8, 256);
}

static int run_nms_stage(const DetectWs& w, int B, long long cap, int K, const BoxSrc& bs, int top_k,
                         float iou_threshold, float score_threshold, float soft_nms_sigma, int max_det,
                         float* det_boxes, float* det_scores, int32_t* det_classes, int32_t* det_index,
                         int32_t* valid, hipStream_t st) {
  NmsParams p;
  p.B = B; p.K = K; p.max_det = max_det; p.top_k = top_k;
  p.soft = soft_nms_sigma > 0.0f ? 1 : 0;
  // the reference hands NonMaxSuppressionV5 sigma/2 and iou_threshold 1.0 in soft mode
  // (postprocessing_ops.py:448-450); TF then uses scale = -0.5 / soft_nms_sigma.
  p.soft_scale = p.soft ? -0.5f / (soft_nms_sigma / 2.0f) : 0.0f;
  p.iou_thr = iou_threshold;
  p.score_thr = score_threshold;
  p.cap = cap;
  const long long max_cands = top_k > 0 ? (long long)top_k : cap;
  // soft NMS: one selection per step (soft_nms_kernel) for lists of up to 5 120 candidates — every shipped configuration
  // (pre_nms_top_k = 5 000); the queue loop of nms_per_class_kernel up to RN_SORT_CAP.  RNET_SOFT_NMS_QUEUE=1 forces the latter.
  static const bool queue_env = [] { const char* e = getenv("RNET_SOFT_NMS_QUEUE"); return e && e[0] == '1'; }();
  const bool stepwise = p.soft && max_cands <= (long long)RN_SOFT_NC * RN_SOFT_THREADS && !queue_env;
  const size_t lds = nms_lds_bytes(p.soft ? (stepwise ? 2 : 1) : 0, max_cands, max_det, &p.soft_off, &p.chunk_cap, &p.sel_cap,
                                   &p.part_off);
  // first sorted chunk of a hard-NMS list: 512 keys (round 5, same box: 1024 -> 512 keys batch-1 serving 1.571 -> 1.552 ms,
  // batch 8 3.551 -> 3.546; 256 / 128 keys 1.66 / 3.57 - 3.60: a second chunk — another radix select — too often)
  p.first_chunk = 512 < p.chunk_cap ? 512 : p.chunk_cap;
  if (stepwise) {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)soft_nms_kernel<RN_SOFT_NC, RN_SOFT_THREADS>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((soft_nms_kernel<RN_SOFT_NC, RN_SOFT_THREADS>), dim3(B * K), dim3(RN_SOFT_THREADS), lds, st, p, w.counts,
                       w.keys, bs, w.sel_scores, w.sel_boxes, w.sel_idx);
  } else {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)nms_per_class_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(nms_per_class_kernel, dim3(B * K), dim3(RN_PP_THREADS), lds, st, p, w.counts, w.keys, bs,
                       w.sel_scores, w.sel_boxes, w.sel_idx);
  }
  RN_CHECK_LAUNCH();
  const size_t lds_m = (size_t)RN_MAX_DET * 8 + 1024 + 8 + 32;
  hipLaunchKernelGGL(merge_kernel, dim3(B), dim3(RN_PP_THREADS), lds_m, st, K, max_det, w.sel_scores,
                     w.sel_boxes, w.sel_idx, (float4*)det_boxes, det_scores, det_classes, det_index, valid);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

static int check_nms_args(const char* fn, int B, long long n, int K, int top_k, float soft_sigma, int max_det) {
  RN_CHECK_ARG(B > 0 && n > 0 && K > 0, "%s: bad B=%d n=%lld K=%d", fn, B, n, K);
  RN_CHECK_ARG(n < (1ll << 31) && (long long)B * K < (1ll << 31), "%s: problem too large", fn);
  RN_CHECK_ARG(max_det >= 1 && max_det <= RN_MAX_DET, "%s: max_det=%d outside 1..%d", fn, max_det, RN_MAX_DET);
  RN_CHECK_ARG((long long)K * max_det <= 16384, "%s: K*max_det=%lld > 16384", fn, (long long)K * max_det);
  if (soft_sigma > 0.0f)
    RN_CHECK_ARG((top_k > 0 && top_k <= RN_SORT_CAP) || n <= RN_SORT_CAP,
                 "%s: soft NMS needs 0 < pre_nms_top_k <= %d (or <= %d candidates)", fn, RN_SORT_CAP, RN_SORT_CAP);
  return RN_OK;
}

extern "C" int rn_detect_per_class(const float* const* class_logits, const int64_t* level_offsets,
                                   int num_levels, int B, int K, const float* boxes, int pre_nms_top_k,
                                   float iou_threshold, float score_threshold, float soft_nms_sigma,
                                   int max_det, float* det_boxes, float* det_scores, int32_t* det_classes,
                                   int32_t* valid, void* workspace, size_t workspace_bytes, void* stream) {
  PPLevels lv;
  RN_CHECK_ARG(boxes && det_boxes && det_scores && det_classes && valid, "rn_detect_per_class: null argument");
  RN_CHECK_ARG(pp_fill_levels(lv, class_logits, level_offsets, num_levels, B > 0 ? B : 1, K > 0 ? K : 1) == 0,
               "rn_detect_per_class: bad level table");
  const long long A = lv.off[num_levels];
  int rc = check_nms_args("rn_detect_per_class", B, A, K, pre_nms_top_k, soft_nms_sigma, max_det);
  if (rc) return rc;
  const size_t need = rn_detect_workspace_bytes(B, A, K, max_det);
  if (!workspace || workspace_bytes < need) {
    rn_set_error("rn_detect_per_class: workspace %zu < %zu", workspace_bytes, need);
    return RN_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream;
  DetectWs w = detect_ws_layout(workspace, B, A, K, max_det);
  RN_CHECK_HIP(hipMemsetAsync(w.counts, 0, (size_t)B * K * 4, st));
  CompactTiles ct;
  ct.num_levels = num_levels;
  ct.tile_begin[0] = 0;
  for (int l = 0; l < num_levels; ++l) {
    const long long n_l = lv.off[l + 1] - lv.off[l];
    ct.tiles_per_img[l] = (int)rn_cdiv(n_l, RN_CT_GROUP * RN_CT_ANCHORS);
    ct.tile_begin[l + 1] = ct.tile_begin[l] + B * ct.tiles_per_img[l];
  }
  // sigmoid(x) <= thr is certain when x < logit(thr) - margin (sigmoid is monotone; the margin
  // covers rn_sigmoidf's <= 3 ulp error many times over); thr outside (0,1) disables the skip
  float x_skip = -INFINITY;
  if (score_threshold > 0.0f && score_threshold < 1.0f)
    x_skip = (float)(log((double)score_threshold / (1.0 - (double)score_threshold)) - 1e-3);
  int cap_list = RN_CT_ANCHORS * K + RN_CT_ANCHORS * K / 4;   // survivors' list: one sub-tile at worst + a quarter
  size_t lds_ct = rn_align_up((size_t)cap_list * 8 + 4 + (size_t)K * 8 + 16, 16);
  if (lds_ct > 64 * 1024) {   // many classes: exactly one sub-tile (every sub-tile with a survivor then flushes at once)
    cap_list = RN_CT_ANCHORS * K;
    lds_ct = rn_align_up((size_t)cap_list * 8 + 4 + (size_t)K * 8 + 16, 16);
  }
  RN_CHECK_ARG(K <= 255 && lds_ct <= 64 * 1024, "rn_detect_per_class: K=%d too large for the compaction tile", K);
  hipLaunchKernelGGL(compact_logits_kernel, dim3(ct.tile_begin[num_levels]), dim3(RN_PP_THREADS), lds_ct, st, lv,
                     ct, B, K, score_threshold, x_skip, w.counts, w.keys, A, cap_list);
  RN_CHECK_LAUNCH();
  BoxSrc bs;
  bs.base = (const float4*)boxes;
  bs.img_stride = A; bs.idx_stride = 1; bs.cls_stride = 0;
  return run_nms_stage(w, B, A, K, bs, pre_nms_top_k, iou_threshold, score_threshold, soft_nms_sigma, max_det,
                       det_boxes, det_scores, det_classes, nullptr, valid, st);
}

extern "C" int rn_nms_per_class(const float* cand_scores, const float* cand_boxes, int B, int n, int K,
                                float iou_threshold, float score_threshold, float soft_nms_sigma, int max_det,
                                float* det_boxes, float* det_scores, int32_t* det_classes, int32_t* det_index,
                                int32_t* valid, void* workspace, size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(cand_scores && cand_boxes && det_boxes && det_scores && det_classes && valid,
               "rn_nms_per_class: null argument");
  int rc = check_nms_args("rn_nms_per_class", B, n, K, 0, soft_nms_sigma, max_det);
  if (rc) return rc;
  const size_t need = rn_nms_workspace_bytes(B, n, K, max_det);
  if (!workspace || workspace_bytes < need) {
    rn_set_error("rn_nms_per_class: workspace %zu < %zu", workspace_bytes, need);
    return RN_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream;
  DetectWs w = detect_ws_layout(workspace, B, n, K, max_det);
  RN_CHECK_HIP(hipMemsetAsync(w.counts, 0, (size_t)B * K * 4, st));
  hipLaunchKernelGGL(compact_scores_kernel, dim3(pp_blocks((long long)B * n * K)), dim3(RN_PP_THREADS), 0, st,
                     cand_scores, B, (long long)n, K, score_threshold, 1, w.counts, w.keys, (long long)n);
  RN_CHECK_LAUNCH();
  BoxSrc bs;
  bs.base = (const float4*)cand_boxes;
  bs.img_stride = (long long)n * K; bs.idx_stride = K; bs.cls_stride = 1;
  return run_nms_stage(w, B, n, K, bs, 0, iou_threshold, score_threshold, soft_nms_sigma, max_det, det_boxes,
                       det_scores, det_classes, det_index, valid, st);
}

extern "C" int rn_topk_per_class(const float* scores, int B, int64_t A, int K, int top_k, float* topk_scores,
                                 int32_t* topk_indices, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  RN_CHECK_ARG(scores && topk_scores && topk_indices, "rn_topk_per_class: null argument");
  RN_CHECK_ARG(B > 0 && A > 0 && K > 0 && top_k > 0 && A < (1ll << 31), "rn_topk_per_class: bad shape");
  const size_t need = rn_topk_workspace_bytes(B, A, K);
  if (!workspace || workspace_bytes < need) {
    rn_set_error("rn_topk_per_class: workspace %zu < %zu", workspace_bytes, need);
    return RN_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream;
  int* counts = (int*)workspace;
  unsigned long long* keys = (unsigned long long*)((char*)workspace + rn_align_up((size_t)B * K * 4, 256));
  RN_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)B * K * 4, st));
  hipLaunchKernelGGL(compact_scores_kernel, dim3(pp_blocks((long long)B * A * K)), dim3(RN_PP_THREADS), 0, st,
                     scores, B, (long long)A, K, 0.0f, 0, counts, keys, (long long)A);
  RN_CHECK_LAUNCH();
  const int k_out = top_k < A ? top_k : (int)A;
  const size_t lds = rn_align_up((size_t)RN_SORT_CAP * 8 + 1024 + 8 + 32, 16);
  RN_CHECK_HIP(hipFuncSetAttribute((const void*)topk_emit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
  hipLaunchKernelGGL(topk_emit_kernel, dim3(B * K), dim3(RN_PP_THREADS), lds, st, K, k_out, counts, keys,
                     (long long)A, topk_scores, topk_indices);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
