// rn_pool.hip — K6/K7/K8: max-pool, FPN top-down fusion, BalanceFeatures.  All bf16 NHWC,
// 8 channels (16 bytes) per thread, HBM-bound.
// Reference: retinanet/model/backbone/resnet.py:304-307 (3x3 s2 SAME pool),
//            retinanet/model/neck/fpn_base.py:25-26,63-69 (2x2 pools for P6/P7),
//            retinanet/model/neck/fpn.py:93-98 + model/layers/feature_fusion.py:41-56 +
//            model/layers/nearest_upsampling.py:19-21 (top-down sum fusion + activation),
//            retinanet/model/layers/balance_features.py:19-60.
#include "rn_common.h"

#define POOL_THREADS 256
#define RN_PYR_MAX 8

struct bf8 { float v[8]; };
__device__ __forceinline__ bf8 unpack8(uint4 u) {
  bf8 r;
  r.v[0] = rn_bf16_to_f32((uint16_t)(u.x & 0xffffu)); r.v[1] = rn_bf16_to_f32((uint16_t)(u.x >> 16));
  r.v[2] = rn_bf16_to_f32((uint16_t)(u.y & 0xffffu)); r.v[3] = rn_bf16_to_f32((uint16_t)(u.y >> 16));
  r.v[4] = rn_bf16_to_f32((uint16_t)(u.z & 0xffffu)); r.v[5] = rn_bf16_to_f32((uint16_t)(u.z >> 16));
  r.v[6] = rn_bf16_to_f32((uint16_t)(u.w & 0xffffu)); r.v[7] = rn_bf16_to_f32((uint16_t)(u.w >> 16));
  return r;
}
__device__ __forceinline__ uint4 pack8(const bf8& r) {
  uint4 u;
  u.x = rn_pack_bf16x2(r.v[0], r.v[1]); u.y = rn_pack_bf16x2(r.v[2], r.v[3]);
  u.z = rn_pack_bf16x2(r.v[4], r.v[5]); u.w = rn_pack_bf16x2(r.v[6], r.v[7]);
  return u;
}
// round every lane to bf16 and back (what a materialised bf16 tensor would hold)
__device__ __forceinline__ void round8(bf8& r) {
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = rn_bf16_to_f32(rn_f32_to_bf16(r.v[i]));
}

// (n, y, x, channel group) of a flat element index.  These kernels are latency-bound at serving batch sizes, and six
// 64-bit integer divisions by run-time values (~100 instructions each) were most of a thread's work: batch-1 serving spent
// 25 us in balance_add_kernel.  32-bit divisions whenever the tensor has fewer than 2^31 elements (uniform branch).
typedef RnIdx4 Idx4;   // rn_common.h: rn_decode4 (mask / shift + float-reciprocal divisions below 2^22 pixels)
__device__ __forceinline__ Idx4 decode4(long long t, int C8, int Wl, int Hl, int mode) { return rn_decode4(t, C8, Wl, Hl, mode); }

static int pool_blocks(long long items) {
  long long b = rn_cdiv(items, POOL_THREADS);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- max pool ------------------------------------------------------------------------------
__global__ void __launch_bounds__(POOL_THREADS)
maxpool_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8, int k, int stride,
               int pt, int pl, int Ho, int Wo) {
  const long long total = (long long)N * Ho * Wo * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const Idx4 d = decode4(i, C8, Wo, Ho, rn_decode_mode(total, C8));
    const int c = d.c, ox = d.x, oy = d.y, n = d.n;
    bf8 m;
#pragma unroll
    for (int q = 0; q < 8; ++q) m.v[q] = -INFINITY;
    for (int r = 0; r < k; ++r) {
      const int iy = oy * stride - pt + r;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int s = 0; s < k; ++s) {
        const int ix = ox * stride - pl + s;
        if ((unsigned)ix >= (unsigned)W) continue;
        const bf8 v = unpack8(x[(((long long)n * H + iy) * W + ix) * C8 + c]);
#pragma unroll
        for (int q = 0; q < 8; ++q) m.v[q] = fmaxf(m.v[q], v.v[q]);
      }
    }
    y[i] = pack8(m);
  }
}

extern "C" int rn_maxpool2d_nhwc(const void* x, void* y, int N, int H, int W, int C, int k, int stride,
                                 int pad_top, int pad_left, int Ho, int Wo, void* stream) {
  RN_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && Ho > 0 && Wo > 0,
               "rn_maxpool2d_nhwc: bad argument");
  RN_CHECK_ARG(C % 8 == 0, "rn_maxpool2d_nhwc: C=%d not a multiple of 8", C);
  const long long total = (long long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(maxpool_kernel, dim3(pool_blocks(total)), dim3(POOL_THREADS), 0, (hipStream_t)stream,
                     (const uint4*)x, (uint4*)y, N, H, W, C / 8, k, stride, pad_top, pad_left, Ho, Wo);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- FPN top-down ------------------------------------------------------------------------------
// out[l](y,x) = act(in[l](y,x) + out[l+1](y/2,x/2)), out[L-1] = in[L-1].  The dependency chain is
// purely local, so each thread re-evaluates it from the coarsest level down (<= L-1 extra
// 16-byte reads, all L2 hits) instead of running L-1 dependent launches; every intermediate is
// rounded to bf16 exactly as the materialised tensor would be.
struct Pyramid {
  int L, N, H0, W0, C8, act;
  const uint4* in[RN_PYR_MAX];
  uint4* out[RN_PYR_MAX];
  long long begin[RN_PYR_MAX + 1];
};

// One launch writes the levels [lo, hi): its chains start from level hi — the coarsest input (hi = L - 1) or an output an
// earlier launch finished.  A thread of the finest level re-evaluates L - 1 stages (unpack, add, activation, rounding:
// ~60 VALU instructions each): at training batch sizes the single launch is VALU-bound (109 us for 275 MB at B = 32), so
// the host cuts the pyramid into launches of ONE stage per element there (rn_fpn_topdown).
__global__ void __launch_bounds__(POOL_THREADS) fpn_topdown_kernel(Pyramid p, int lo, int hi) {
  const long long first = p.begin[lo], total = p.begin[hi];
  const uint4* __restrict__ src_hi = hi == p.L - 1 ? p.in[hi] : p.out[hi];
  for (long long i = first + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = lo;
    while (i >= p.begin[l + 1]) ++l;
    const int Wl = p.W0 >> l, Hl = p.H0 >> l;
    const Idx4 d = decode4(i - p.begin[l], p.C8, Wl, Hl, rn_decode_mode(total, p.C8));
    const int c = d.c, x = d.x, y = d.y, n = d.n;
    bf8 v = unpack8(src_hi[(((long long)n * (p.H0 >> hi)) + (y >> (hi - l))) * (p.W0 >> hi) * p.C8 +
                           (long long)(x >> (hi - l)) * p.C8 + c]);
    for (int k = hi - 1; k >= l; --k) {
      const int Hk = p.H0 >> k, Wk = p.W0 >> k;
      const bf8 u = unpack8(p.in[k][(((long long)n * Hk) + (y >> (k - l))) * Wk * p.C8 +
                                   (long long)(x >> (k - l)) * p.C8 + c]);
#pragma unroll
      for (int q = 0; q < 8; ++q) v.v[q] = u.v[q] + v.v[q];
      rn_apply_act_n<8>(v.v, p.act);
      round8(v);
    }
    p.out[l][(((long long)n * Hl) + y) * Wl * p.C8 + (long long)x * p.C8 + c] = pack8(v);
  }
}

static int fill_pyramid(Pyramid& p, void* const* in, void* const* out, int L, int N, int H0, int W0, int C) {
  if (L < 2 || L > RN_PYR_MAX || N <= 0 || C % 8 != 0 || !in || !out) return -1;
  if ((H0 >> (L - 1)) < 1 || (W0 >> (L - 1)) < 1) return -1;
  if ((H0 % (1 << (L - 1))) || (W0 % (1 << (L - 1)))) return -1;
  p.L = L; p.N = N; p.H0 = H0; p.W0 = W0; p.C8 = C / 8;
  p.begin[0] = 0;
  for (int l = 0; l < L; ++l) {
    if (!in[l] || !out[l]) return -1;
    p.in[l] = (const uint4*)in[l];
    p.out[l] = (uint4*)out[l];
    p.begin[l + 1] = p.begin[l] + (long long)N * (H0 >> l) * (W0 >> l) * (C / 8);
  }
  return 0;
}

extern "C" int rn_fpn_topdown(void* const* p_in, void* const* p_out, int num_levels, int N, int H0, int W0,
                              int C, int act, void* stream) {
  Pyramid p;
  RN_CHECK_ARG(fill_pyramid(p, p_in, p_out, num_levels, N, H0, W0, C) == 0,
               "rn_fpn_topdown: bad pyramid (levels must halve exactly, C %% 8 == 0)");
  p.act = act;
  // small pyramids (serving): one launch, every thread walks its chain from the coarsest level.  Large ones: the two
  // finest levels (80 of every 85 elements of a five-level pyramid) as launches of ONE stage per element, behind a first
  // launch for the coarse rest
  int cuts[4] = {num_levels - 1, 0, 0, 0}, ncuts = 1;
  if (num_levels >= 3 && p.begin[1] >= (1ll << 21)) {
    ncuts = 0;
    if (num_levels > 3) cuts[ncuts++] = num_levels - 1;   // levels [2, L - 1) from the coarsest input
    cuts[ncuts++] = 2;
    cuts[ncuts++] = 1;
  }
  for (int q = 0; q < ncuts; ++q) {
    const int hi = cuts[q], lo = q + 1 < ncuts ? cuts[q + 1] : 0;
    hipLaunchKernelGGL(fpn_topdown_kernel, dim3(pool_blocks(p.begin[hi] - p.begin[lo])), dim3(POOL_THREADS), 0,
                       (hipStream_t)stream, p, lo, hi);
    RN_CHECK_LAUNCH();
  }
  return RN_OK;
}

// ---- BalanceFeatures -----------------------------------------------------------------------------
// pass 1: avg(y,x) at the intermediate level = mean over levels of {max-pooled finer levels,
// nearest-upsampled coarser levels} (balance_features.py:23-40); pass 2: every level adds the
// average resized back to its own resolution (:42-58).
struct Balance {
  Pyramid p;
  int mid;
  uint4* avg;
};

__global__ void __launch_bounds__(POOL_THREADS) balance_avg_kernel(Balance b) {
  const Pyramid& p = b.p;
  const int Hm = p.H0 >> b.mid, Wm = p.W0 >> b.mid;
  const long long total = (long long)p.N * Hm * Wm * p.C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const Idx4 d = decode4(i, p.C8, Wm, Hm, rn_decode_mode(total, p.C8));
    const int c = d.c, x = d.x, y = d.y, n = d.n;
    bf8 acc;
#pragma unroll
    for (int q = 0; q < 8; ++q) acc.v[q] = 0.0f;
    for (int l = 0; l < p.L; ++l) {
      const int Hl = p.H0 >> l, Wl = p.W0 >> l;
      bf8 v;
      if (l >= b.mid) {
        const int sh = l - b.mid;
        v = unpack8(p.in[l][(((long long)n * Hl) + (y >> sh)) * Wl * p.C8 + (long long)(x >> sh) * p.C8 + c]);
      } else {
        const int f = 1 << (b.mid - l);
#pragma unroll
        for (int q = 0; q < 8; ++q) v.v[q] = -INFINITY;
        for (int dy = 0; dy < f; ++dy)
          for (int dx = 0; dx < f; ++dx) {
            const bf8 u = unpack8(
                p.in[l][(((long long)n * Hl) + (y * f + dy)) * Wl * p.C8 + (long long)(x * f + dx) * p.C8 + c]);
#pragma unroll
            for (int q = 0; q < 8; ++q) v.v[q] = fmaxf(v.v[q], u.v[q]);
          }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) acc.v[q] += v.v[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc.v[q] = acc.v[q] / (float)p.L;
    b.avg[i] = pack8(acc);
  }
}

__global__ void __launch_bounds__(POOL_THREADS) balance_add_kernel(Balance b) {
  const Pyramid& p = b.p;
  const int Hm = p.H0 >> b.mid, Wm = p.W0 >> b.mid;
  const long long total = p.begin[p.L];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = 0;
    while (i >= p.begin[l + 1]) ++l;
    const int Wl = p.W0 >> l, Hl = p.H0 >> l;
    const Idx4 d = decode4(i - p.begin[l], p.C8, Wl, Hl, rn_decode_mode(total, p.C8));
    const int c = d.c, x = d.x, y = d.y, n = d.n;
    bf8 a;
    if (l <= b.mid) {
      const int sh = b.mid - l;
      a = unpack8(b.avg[(((long long)n * Hm) + (y >> sh)) * Wm * p.C8 + (long long)(x >> sh) * p.C8 + c]);
    } else {
      const int f = 1 << (l - b.mid);
#pragma unroll
      for (int q = 0; q < 8; ++q) a.v[q] = -INFINITY;
      // A coarse-level thread pools f x f values of the average (64 for the coarsest level of a five-level pyramid).  As a
      // rolled loop that is 64 dependent L2 round trips — 25 us of batch-1 serving, whatever the tensor sizes; in rows of
      // up to eight independent loads the latency is paid f times.
      const uint4* row0 = b.avg + (((long long)n * Hm) + (long long)y * f) * Wm * p.C8 + (long long)(x * f) * p.C8 + c;
      for (int dy = 0; dy < f; ++dy) {
        const uint4* rp = row0 + (long long)dy * Wm * p.C8;
        for (int dx0 = 0; dx0 < f; dx0 += 8) {
          uint4 raw[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)   // f is a power of two: below 8 the extra lanes re-read a column of the window (max unchanged)
            raw[j] = rp[(long long)(f >= 8 ? dx0 + j : (j & (f - 1))) * p.C8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bf8 u = unpack8(raw[j]);
#pragma unroll
            for (int q = 0; q < 8; ++q) a.v[q] = fmaxf(a.v[q], u.v[q]);
          }
        }
      }
    }
    const long long o = (((long long)n * Hl) + y) * Wl * p.C8 + (long long)x * p.C8 + c;
    const bf8 f = unpack8(p.in[l][o]);
#pragma unroll
    for (int q = 0; q < 8; ++q) a.v[q] = f.v[q] + a.v[q];
    p.out[l][o] = pack8(a);
  }
}

extern "C" int rn_balance_features(void* const* p_in, void* const* p_out, int num_levels, int mid, int N,
                                   int H0, int W0, int C, void* scratch, void* stream) {
  Balance b;
  RN_CHECK_ARG(fill_pyramid(b.p, p_in, p_out, num_levels, N, H0, W0, C) == 0,
               "rn_balance_features: bad pyramid (levels must halve exactly, C %% 8 == 0)");
  RN_CHECK_ARG(mid >= 0 && mid < num_levels && scratch, "rn_balance_features: bad mid/scratch");
  b.p.act = 0;
  b.mid = mid;
  b.avg = (uint4*)scratch;
  const long long nm = (long long)N * (H0 >> mid) * (W0 >> mid) * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(balance_avg_kernel, dim3(pool_blocks(nm)), dim3(POOL_THREADS), 0, st, b);
  RN_CHECK_LAUNCH();
  hipLaunchKernelGGL(balance_add_kernel, dim3(pool_blocks(b.p.begin[num_levels])), dim3(POOL_THREADS), 0, st, b);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
