// rn_depthwise.hip — a18: the EfficientNet / SeparableConv2D building blocks that are not GEMMs.
//   * depthwise k x k convolution (tf.keras.layers.DepthwiseConv2D in MBConvBlock,
//     retinanet/model/backbone/efficientnet.py:372-380, and the depthwise half of SeparableConv2D
//     in fpn_base.py:28-39 / detection_head.py:37-50), TF SAME padding, fused BN scale/shift + act;
//   * squeeze-and-excitation (efficientnet.py:222-265): global average pool, the two 1x1 "convs" on
//     the pooled vector (swish, sigmoid) and the per-(image, channel) gate.
// 16-bit NHWC (rn_common.h: bfloat16, or IEEE half in the -DRN_F16 build), 8 channels (16 B) per thread, fp32 math.
// HBM-bound in principle; the k x k tap loops are VALU-issue bound in practice (packed fp32 pairs, rolled filter rows:
// see depthwise_strip_kernel).  Grouped like the GEMM convs so the five pyramid levels of a shared separable head conv
// go out in one launch.
#include "rn_common.h"

#define DW_THREADS 256

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
// Packed fp32 pairs (two lanes of fp32 per VALU slot).  The depthwise kernels are VALU-bound — a 5x5 tap loop is ~530
// instructions per 16-byte output with scalar fp32 — and the packed forms halve the count.  Round 6: the forward / data-
// gradient tap loop uses v_pk_fma_f32 (one rounding per tap instead of two: the library is built with -ffp-contract=off
// for the bit-exact integer / loss kernels, so the fused form has to be asked for; TF's DepthwiseConv2D fixes no order of
// operations and the parity test is a tolerance against an fp32 convolution) — another 80 of ~260 instructions per filter
// row of a 5x5 strip.
typedef float dw_f2 __attribute__((ext_vector_type(2)));
struct bf8p { dw_f2 v[4]; };
__device__ __forceinline__ bf8p unpack8p(uint4 u) {
  bf8p r;
  r.v[0] = dw_f2{rn_lo16(u.x), rn_hi16(u.x)};
  r.v[1] = dw_f2{rn_lo16(u.y), rn_hi16(u.y)};
  r.v[2] = dw_f2{rn_lo16(u.z), rn_hi16(u.z)};
  r.v[3] = dw_f2{rn_lo16(u.w), rn_hi16(u.w)};
  return r;
}
// per-element form on purpose: with the array form (rn_apply_act_n) the 5x5 strip kernel's unrolled 4-pixel
// epilogue compiled 4.5x slower (361 us against 79 us per launch on EfficientNet-B3)
__device__ __forceinline__ float act_exact(float v, int act) {
  switch (act) {
    case RN_ACT_RELU: return fmaxf(v, 0.0f);
    case RN_ACT_RELU6: return fminf(fmaxf(v, 0.0f), 6.0f);
    case RN_ACT_SWISH: return rn_swish(v);
    default: return v;
  }
}

struct DwSegDev {
  const uint4* x; const uint4* w; uint4* y; const float* scale; const float* shift; const uint4* residual;
  int N, H, W, C8, Ho, Wo;
  long long begin;
};
struct DwArgs {
  int k, stride, pt, pl, act, nseg;
  long long total;
  DwSegDev seg[RN_CONV_MAX_SEGMENTS];
};

// Strip kernel: one thread = T = 4 consecutive output pixels of one row x 8 channels.  Per filter row the
// (T-1)*stride + K input vectors are loaded once (all in flight together) and reused by every tap that
// touches them: K*((T-1)*S+K) 16-byte loads per strip instead of T*K*K, fully unrolled.  (The first version
// looped over the taps with data-dependent `continue`s: one L1/L2 round trip after the other, ~10x off HBM.)
// EPI: the epilogue's shape, decided on the host and compiled in (bit 0: scale / shift, bit 1: residual input, bit 2:
// swish; EPI_RUNTIME: the segments disagree, test at run time).  With the tests at run time every path's registers
// count against a kernel that already sits at the 256-VGPR limit (K = 5: 780 bytes of scratch per lane and twice the
// time, K = 3: 216 -> 252 VGPRs) — the rounding points added in round 2 did exactly that.
constexpr int EPI_RUNTIME = 8;
template <int K, int S, int EPI>
__global__ void __launch_bounds__(DW_THREADS) depthwise_strip_kernel(const DwArgs a) {
  constexpr int T = 4;
  constexpr int WIN = (T - 1) * S + K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.total;
       i += (long long)gridDim.x * blockDim.x) {
    int si = 0;
    while (si + 1 < a.nseg && i >= a.seg[si + 1].begin) ++si;
    const DwSegDev& s = a.seg[si];
    const unsigned strips = (unsigned)(s.Wo + T - 1) / T;
    unsigned t = (unsigned)(i - s.begin);          // < 2^31 per segment (host check)
    const int c = (int)(t % (unsigned)s.C8);
    t /= (unsigned)s.C8;
    const int ox0 = (int)(t % strips) * T;
    t /= strips;
    const int oy = (int)(t % (unsigned)s.Ho);
    const int n = (int)(t / (unsigned)s.Ho);
    dw_f2 acc[T][4];
#pragma unroll
    for (int tt = 0; tt < T; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[tt][q] = dw_f2{0.0f, 0.0f};
    const int ix0 = ox0 * S - a.pl;
    // ROLLED over the filter rows: unrolled, the compiler keeps several rows' windows, weights and their unpacked
    // floats live — 256 VGPRs + ~200 AGPRs for K = 5, one wave per SIMD — and a wave then sits out every L2 round trip
    // alone.  Rolled: 126-166 VGPRs, 3-4 waves per SIMD; the 5x5 layers of EfficientNet-B3 run 1.9-2.3x faster
    // (k5 / 80x80 / 288 ch, batch 32: 313 -> 156 us forward, 269 -> 116 us as a data gradient), the 3x3 ones 1.1-1.3x.
    // (An LDS-staged variant — rows DMA'd once into a ring, fragments read from LDS — was built and measured at 198 us
    // for the same layer: the kernel is bound by VALU issue and occupancy, not by where the window comes from.)
#pragma unroll 1
    for (int r = 0; r < K; ++r) {
      const int iy = oy * S - a.pt + r;
      const bool rowok = (unsigned)iy < (unsigned)s.H;
      const uint4* xrow = s.x + ((long long)n * s.H + (rowok ? iy : 0)) * s.W * s.C8 + c;
      uint4 win[WIN];
#pragma unroll
      for (int j = 0; j < WIN; ++j) {
        const int ix = ix0 + j;
        win[j] = (rowok && (unsigned)ix < (unsigned)s.W) ? xrow[(long long)ix * s.C8] : make_uint4(0u, 0u, 0u, 0u);
      }
      uint4 wr[K];
#pragma unroll
      for (int ss = 0; ss < K; ++ss) wr[ss] = s.w[(long long)(r * K + ss) * s.C8 + c];
#pragma unroll
      for (int j = 0; j < WIN; ++j) {
        const bf8p xv = unpack8p(win[j]);
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
          const int ss = j - tt * S;          // compile-time after unrolling
          if (ss >= 0 && ss < K) {
            const bf8p wv = unpack8p(wr[ss]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[tt][q] = __builtin_elementwise_fma(xv.v[q], wv.v[q], acc[tt][q]);   // v_pk_fma_f32
          }
        }
      }
    }
    const bool affine = EPI == EPI_RUNTIME ? (s.scale != nullptr || s.shift != nullptr) : (EPI & 1) != 0;
    const bool has_res = EPI == EPI_RUNTIME ? s.residual != nullptr : (EPI & 2) != 0;
    const bool swish = EPI == EPI_RUNTIME ? a.act == RN_ACT_SWISH : (EPI & 4) != 0;
    float sc[8], sh[8];
    if (affine) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        sc[q] = s.scale ? s.scale[c * 8 + q] : 1.0f;
        sh[q] = s.shift ? s.shift[c * 8 + q] : 0.0f;
      }
    }
#pragma unroll
    for (int tt = 0; tt < T; ++tt) {
      if (ox0 + tt >= s.Wo) break;
      const long long oi = (((long long)n * s.Ho + oy) * s.Wo + ox0 + tt) * s.C8 + c;
      bf8 o, res;
      if (has_res) res = unpack8(s.residual[oi]);
      // bf16 tensors where the reference has them: DepthwiseConv2D output, BatchNorm output in front of swish
      // (rnet_hip.h, rn_conv_segment); the accumulate form (data gradients: no affine) adds in fp32, one rounding
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v = acc[tt][q >> 1][q & 1];
        if (affine) v = rn_rb(v) * sc[q] + sh[q];
        if (has_res) v = (affine ? rn_rb(v) : v) + res.v[q];
        if (swish) v = rn_rb(v);
        o.v[q] = swish ? act_exact(v, RN_ACT_SWISH) : act_exact(v, a.act);
      }
      s.y[oi] = pack8(o);
    }
  }
}

extern "C" int rn_depthwise_conv2d_nhwc_fwd(const rn_dw_problem* p, void* stream) {
  RN_CHECK_ARG(p && p->num_segments >= 1 && p->num_segments <= RN_CONV_MAX_SEGMENTS &&
                   (p->k == 1 || p->k == 3 || p->k == 5) && (p->stride == 1 || p->stride == 2),
               "rn_depthwise_conv2d_nhwc_fwd: bad problem (k in {1,3,5}, stride in {1,2})");
  DwArgs a;
  a.k = p->k; a.stride = p->stride; a.pt = p->pad_top; a.pl = p->pad_left; a.act = p->act; a.nseg = p->num_segments;
  long long off = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_dw_segment& s = p->seg[i];
    RN_CHECK_ARG(s.x && s.w && s.y && s.C % 8 == 0 && s.N > 0 && s.H > 0 && s.W > 0 && s.Ho > 0 && s.Wo > 0,
                 "rn_depthwise_conv2d_nhwc_fwd: segment %d bad (C %% 8 == 0)", i);
    DwSegDev& d = a.seg[i];
    d.x = (const uint4*)s.x; d.w = (const uint4*)s.w; d.y = (uint4*)s.y; d.scale = s.scale; d.shift = s.shift;
    d.residual = (const uint4*)s.residual;
    d.N = s.N; d.H = s.H; d.W = s.W; d.C8 = s.C / 8; d.Ho = s.Ho; d.Wo = s.Wo;
    d.begin = off;
    RN_CHECK_ARG((long long)s.N * s.Ho * s.Wo * (s.C / 8) < (1ll << 31) && (long long)s.N * s.H * s.W * (s.C / 8) < (1ll << 31),
                 "rn_depthwise_conv2d_nhwc_fwd: segment %d too large", i);
    off += (long long)s.N * s.Ho * ((s.Wo + 3) / 4) * (s.C / 8);   // work items: 4-pixel strips x 8-channel groups
  }
  a.total = off;
  long long blocks = rn_cdiv(off, DW_THREADS);
  if (blocks > 32768) blocks = 32768;
  const dim3 grid((unsigned)blocks), block(DW_THREADS);
  hipStream_t st = (hipStream_t)stream;
  // the epilogue variant: what every segment agrees on (forward: affine + swish; data gradients: plain or accumulate)
  int epi = ((p->seg[0].scale || p->seg[0].shift) ? 1 : 0) | (p->seg[0].residual ? 2 : 0) | (p->act == RN_ACT_SWISH ? 4 : 0);
  for (int i = 1; i < p->num_segments; ++i) {
    const rn_dw_segment& s = p->seg[i];
    if ((((s.scale || s.shift) ? 1 : 0) | (s.residual ? 2 : 0)) != (epi & 3)) epi = EPI_RUNTIME;
  }
  const int key = p->k * 10 + p->stride;
#define DW_LAUNCH_(K_, S_)                                                                                          \
  switch (epi) {                                                                                                    \
    case 0: hipLaunchKernelGGL((depthwise_strip_kernel<K_, S_, 0>), grid, block, 0, st, a); break;                  \
    case 1: hipLaunchKernelGGL((depthwise_strip_kernel<K_, S_, 1>), grid, block, 0, st, a); break;                  \
    case 2: hipLaunchKernelGGL((depthwise_strip_kernel<K_, S_, 2>), grid, block, 0, st, a); break;                  \
    case 5: hipLaunchKernelGGL((depthwise_strip_kernel<K_, S_, 5>), grid, block, 0, st, a); break;                  \
    default: hipLaunchKernelGGL((depthwise_strip_kernel<K_, S_, EPI_RUNTIME>), grid, block, 0, st, a); break;       \
  }
  switch (key) {
    case 11: DW_LAUNCH_(1, 1); break;
    case 12: DW_LAUNCH_(1, 2); break;
    case 31: DW_LAUNCH_(3, 1); break;
    case 32: DW_LAUNCH_(3, 2); break;
    case 51: DW_LAUNCH_(5, 1); break;
    default: DW_LAUNCH_(5, 2); break;
  }
#undef DW_LAUNCH_
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// HWC1 f32 depthwise kernel [k,k,C,1] -> bf16 [k*k][C]
__global__ void pack_dw_kernel(const float* __restrict__ w, long long n, uint16_t* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = rn_f32_to_bf16(w[i]);
}
extern "C" int rn_pack_depthwise_weight(const float* w, int k, int C, void* out, void* stream) {
  RN_CHECK_ARG(w && out && k > 0 && C > 0, "rn_pack_depthwise_weight: bad argument");
  const long long n = (long long)k * k * C;
  hipLaunchKernelGGL(pack_dw_kernel, dim3((unsigned)rn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, n,
                     (uint16_t*)out);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// master f32 [k*k][C] -> bf16 [k*k][C] with the taps reversed (the filter of the data gradient)
__global__ void pack_dw_flip_kernel(const float* __restrict__ w, int taps, int C, uint16_t* __restrict__ out) {
  const long long n = (long long)taps * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i / C), c = (int)(i - (long long)t * C);
    out[i] = rn_f32_to_bf16(w[(long long)(taps - 1 - t) * C + c]);
  }
}
extern "C" int rn_pack_depthwise_weight_flip(const float* w, int k, int C, void* out, void* stream) {
  RN_CHECK_ARG(w && out && k > 0 && C > 0, "rn_pack_depthwise_weight_flip: bad argument");
  const long long n = (long long)k * k * C;
  hipLaunchKernelGGL(pack_dw_flip_kernel, dim3((unsigned)rn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     k * k, C, (uint16_t*)out);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- depthwise weight gradient --------------------------------------------------------------------
// dW[tap][c] = sum over output pixels of dy[p][c] * x[p shifted by tap][c]; two-stage deterministic
// reduction: (pixel chunk, channel slab) workgroups -> partials[chunk][tap][C] -> ordered sum.
// Thread = CPT channels x one pixel lane (32 lanes), all K*K taps in registers.
struct DwgSegDev {
  const uint16_t* x; const uint16_t* dy;
  int N, H, W, Ho, Wo, P, chunk_begin;
};
struct DwgArgs {
  int stride, pt, pl, nseg, C, rows_per_chunk, total_chunks;
  int slab_groups;   // channel groups (CPT channels each) per workgroup: min(C / CPT, 64)
  float* partial;
  DwgSegDev seg[RN_CONV_MAX_SEGMENTS];
};

// the same as packed pairs (v_pk_mul_f32 / v_pk_add_f32 in the tap loop)
template <int CPT>
__device__ __forceinline__ void dwg_load_p(const uint16_t* p, dw_f2* v) {
  if (CPT == 8) {
    const uint4 u = *(const uint4*)p;
    v[0] = dw_f2{rn_lo16(u.x), rn_hi16(u.x)};
    v[1] = dw_f2{rn_lo16(u.y), rn_hi16(u.y)};
    v[2] = dw_f2{rn_lo16(u.z), rn_hi16(u.z)};
    v[3] = dw_f2{rn_lo16(u.w), rn_hi16(u.w)};
  } else {
    const uint2 u = *(const uint2*)p;
    v[0] = dw_f2{rn_lo16(u.x), rn_hi16(u.x)};
    v[1] = dw_f2{rn_lo16(u.y), rn_hi16(u.y)};
  }
}
// the same through a buffer descriptor: an offset past the descriptor's size (DWG_OOB) returns zeros — the hardware's
// bounds check instead of a branch around the load.  Tensors up to 4 GB - 64 B (32-bit record count / offsets).
#define DWG_OOB 0xfffffff0u
typedef unsigned dwg_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned dwg_u32x2 __attribute__((ext_vector_type(2)));
template <int CPT>
__device__ __forceinline__ void dwg_bload(__amdgpu_buffer_rsrc_t rs, unsigned off, dw_f2* v) {
  if (CPT == 8) {
    const dwg_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    v[0] = dw_f2{rn_lo16(u.x), rn_hi16(u.x)};
    v[1] = dw_f2{rn_lo16(u.y), rn_hi16(u.y)};
    v[2] = dw_f2{rn_lo16(u.z), rn_hi16(u.z)};
    v[3] = dw_f2{rn_lo16(u.w), rn_hi16(u.w)};
  } else {
    const dwg_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    v[0] = dw_f2{rn_lo16(u.x), rn_hi16(u.x)};
    v[1] = dw_f2{rn_lo16(u.y), rn_hi16(u.y)};
  }
}
// CPT consecutive bf16 channels with ONE 16-byte (CPT = 8) or 8-byte (CPT = 4) load
template <int CPT>
__device__ __forceinline__ void dwg_load(const uint16_t* p, float* v) {
  if (CPT == 8) {
    const uint4 u = *(const uint4*)p;
    v[0] = rn_bf16_to_f32((uint16_t)(u.x & 0xffffu)); v[1] = rn_bf16_to_f32((uint16_t)(u.x >> 16));
    v[2] = rn_bf16_to_f32((uint16_t)(u.y & 0xffffu)); v[3] = rn_bf16_to_f32((uint16_t)(u.y >> 16));
    v[4] = rn_bf16_to_f32((uint16_t)(u.z & 0xffffu)); v[5] = rn_bf16_to_f32((uint16_t)(u.z >> 16));
    v[6] = rn_bf16_to_f32((uint16_t)(u.w & 0xffffu)); v[7] = rn_bf16_to_f32((uint16_t)(u.w >> 16));
  } else {
    const uint2 u = *(const uint2*)p;
    v[0] = rn_bf16_to_f32((uint16_t)(u.x & 0xffffu)); v[1] = rn_bf16_to_f32((uint16_t)(u.x >> 16));
    v[2] = rn_bf16_to_f32((uint16_t)(u.y & 0xffffu)); v[3] = rn_bf16_to_f32((uint16_t)(u.y >> 16));
  }
}

// Work item = one strip of T = 4 consecutive output pixels of an image row; a chunk = rows_per_chunk items.
// Per filter row the strip's (T-1)*S + K input vectors are loaded once and feed every (pixel, tap) pair that
// touches them (the first version re-read x for each of the K*K taps: L2-bandwidth bound).
//
// Thread mapping (round 6).  A thread owns CPT consecutive channels — its K*K x CPT accumulators — and walks items; the
// threads of a workgroup are (channel group cg, item lane rl).  Until round 5 a workgroup covered a fixed 64-channel slab
// (8 groups x 32 item lanes): with EfficientNet's channel counts — 144, 288, 816, 1392: none a multiple of 64 — a pixel's
// slab was a 128-byte piece at an odd offset of a 288-byte pixel (every piece straddles two cache lines, so the launch
// fetched ~1.8 x its bytes) and the last slab ran with most of its channel lanes dead (144 = 64 + 64 + 16).  Now the
// channel groups of a workgroup are min(C / CPT, 64) — the WHOLE pixel whenever it has at most 64 groups, so consecutive
// lanes read consecutive bytes across pixel boundaries — and the 256 threads are cut into 256 / groups item lanes.
template <int K, int CPT, int S>
__global__ void __launch_bounds__(256) depthwise_wgrad_kernel(const DwgArgs a) {
  constexpr int T = 4;
  constexpr int WIN = (T - 1) * S + K;
  const int chunk = blockIdx.x, slab = blockIdx.y;
  int si = 0;
  while (si + 1 < a.nseg && chunk >= a.seg[si + 1].chunk_begin) ++si;
  const DwgSegDev& s = a.seg[si];
  const int CG = a.C / CPT;                                   // channel groups of a pixel
  const int g0 = slab * a.slab_groups;                         // first group of this workgroup
  const int SG = CG - g0 < a.slab_groups ? CG - g0 : a.slab_groups;   // its groups
  const int RL = 256 / SG;                                     // item lanes
  const int cg = threadIdx.x % SG, rl = threadIdx.x / SG;
  const int c0 = (g0 + cg) * CPT;
  const bool live = rl < RL;
  constexpr int CP2 = CPT / 2;
  dw_f2 acc[K * K][CP2];
#pragma unroll
  for (int t = 0; t < K * K; ++t)
#pragma unroll
    for (int q = 0; q < CP2; ++q) acc[t][q] = dw_f2{0.0f, 0.0f};
  const int strips = (s.Wo + T - 1) / T;
  const int items = s.N * s.Ho * strips;           // s.P holds the item count of the segment
  const int i0 = (chunk - s.chunk_begin) * a.rows_per_chunk;
  const int i1 = i0 + a.rows_per_chunk < items ? i0 + a.rows_per_chunk : items;
  // (tensors below 4 GB: checked by the host; the record count is an unsigned 32-bit field)
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)s.x, 0, (int)(unsigned)((long long)s.N * s.H * s.W * a.C * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc((void*)s.dy, 0, (int)(unsigned)((long long)s.N * s.Ho * s.Wo * a.C * 2), 0x00020000);
  if (live) {
    for (int it = i0 + rl; it < i1; it += RL) {
      const int ox0 = (it % strips) * T;
      const int t2 = it / strips;
      const int oy = t2 % s.Ho;
      const int n = t2 / s.Ho;
      // Every load of an item is UNCONDITIONAL: a buffer load whose offset is pushed out of range for a pixel outside the
      // image, so the hardware returns zeros.  With the bounds tests as branches around the loads (until round 6) the
      // compiler emitted `load; s_waitcnt vmcnt(0); branch` per load: 22 (K = 3) to 44 (K = 5) DEPENDENT L2 round trips
      // per item, which is why these launches ran at 0.5 - 1.5 TB/s.
      dw_f2 g[T][CP2];
      const unsigned dyoff = (unsigned)(((((long long)n * s.Ho + oy) * s.Wo + ox0) * a.C + c0) * 2);
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
        dwg_bload<CPT>(rs_dy, ox0 + tt < s.Wo ? dyoff + (unsigned)(tt * a.C * 2) : DWG_OOB, g[tt]);
      const int ix0 = ox0 * S - a.pl;
#pragma unroll
      for (int r = 0; r < K; ++r) {
        const int iy = oy * S - a.pt + r;
        const bool rok = (unsigned)iy < (unsigned)s.H;
        const unsigned xoff = (unsigned)(((((long long)n * s.H + iy) * s.W + ix0) * a.C + c0) * 2);   // (garbage when !rok: unused)
        dw_f2 xv[WIN][CP2];
#pragma unroll
        for (int j = 0; j < WIN; ++j)
          dwg_bload<CPT>(rs_x, rok && (unsigned)(ix0 + j) < (unsigned)s.W ? xoff + (unsigned)(j * a.C * 2) : DWG_OOB, xv[j]);
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
#pragma unroll
          for (int tt = 0; tt < T; ++tt) {
            const int ss = j - tt * S;     // compile-time after unrolling
            if (ss >= 0 && ss < K) {
#pragma unroll
              for (int q = 0; q < CP2; ++q) acc[r * K + ss][q] = __builtin_elementwise_fma(g[tt][q], xv[j][q], acc[r * K + ss][q]);
            }
          }
        }
      }
    }
  }
  // item lanes -> one partial row per tap.  K taps (one filter row) per round through red[tap][item lane][channel of the
  // workgroup] (pitch SG * CPT + 1: bank spread): K rounds of two barriers where there were K * K, and the sums over the
  // item lanes are spread over all 256 threads (tap, channel) instead of one thread per channel — with short chunks the
  // old epilogue cost as much as the item loop.
  constexpr int PLANE = 256 * CPT + 256;
  __shared__ float red[K * PLANE];
  const int pitch = SG * CPT + 1;
  const int nch = SG * CPT;
  // fully unrolled: with a run-time tap index the accumulators are an indexed private array, i.e. they live in scratch
  // memory for the whole kernel (304 / 416 bytes per lane) and every accumulate goes through it
#pragma unroll
  for (int r = 0; r < K; ++r) {
    if (live) {
#pragma unroll
      for (int ss = 0; ss < K; ++ss)
#pragma unroll
        for (int q = 0; q < CPT; ++q) red[ss * PLANE + rl * pitch + cg * CPT + q] = acc[r * K + ss][q >> 1][q & 1];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < K * nch; o += 256) {
      const int ss = o / nch, cl = o - ss * nch;
      const float* col = red + ss * PLANE + cl;
      float sum = 0.0f;
      for (int q = 0; q < RL; ++q) sum += col[q * pitch];
      a.partial[((long long)chunk * (K * K) + r * K + ss) * a.C + g0 * CPT + cl] = sum;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256)
dwg_reduce_kernel(const float* __restrict__ partial, long long n, int chunks, float* __restrict__ dw) {
  // 16 outputs x 16 chunk lanes per workgroup: lane j adds chunks j, j + 16, ... in order, the 16 lane sums are then
  // added in lane order — a fixed association (deterministic).  One thread per output walking all ~512 chunks was a
  // 512-deep dependent load chain on a few dozen workgroups: 144 us per launch, 6.5 ms per EfficientNet-B3 step.
  __shared__ float red[16][17];
  const int ol = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const long long i = blockIdx.x * 16ll + ol;
  float t = 0.0f;
  if (i < n)
    for (int c = lane; c < chunks; c += 16) t += partial[(long long)c * n + i];
  red[lane][ol] = t;
  __syncthreads();
  if (threadIdx.x < 16 && i < n) {
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += red[j][ol];
    dw[i] = sum;
  }
}

static int dwg_plan(const rn_dw_problem* p, DwgArgs& a) {
  if (!p || p->num_segments < 1 || p->num_segments > RN_CONV_MAX_SEGMENTS) return -1;
  if (p->k != 1 && p->k != 3 && p->k != 5) return -1;
  a.stride = p->stride; a.pt = p->pad_top; a.pl = p->pad_left; a.nseg = p->num_segments; a.C = p->seg[0].C;
  if (a.C % 8) return -1;
  if (p->stride != 1 && p->stride != 2) return -1;
  long long Ptot = 0;                            // work items: 4-pixel strips of output rows
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_dw_segment& s = p->seg[i];
    if (!s.x || !s.y || s.C != a.C) return -1;
    if ((long long)s.N * s.H * s.W * s.C * 2 >= (1ll << 32) - 64 || (long long)s.N * s.Ho * s.Wo * s.C * 2 >= (1ll << 32) - 64)
      return -1;   // the kernel addresses x / dy through 32-bit buffer offsets
    Ptot += (long long)s.N * s.Ho * ((s.Wo + 3) / 4);
  }
  // ~512 item chunks over the launch, as before (each chunk's workgroups write K*K*C partial sums and run the reduction
  // epilogue: 2048 chunks measured 10 - 30 % slower than 512 although they fill the chip better)
  const int cpt = p->k == 5 ? 4 : 8;             // channels per thread (25 taps x 8 channels would not fit the registers)
  const int cg = a.C / cpt;
  a.slab_groups = cg < 64 ? cg : 64;
  const long long target = 512;
  long long rows = rn_cdiv(Ptot, target);
  rows = rn_cdiv(rows, 32) * 32;
  if (rows < 32) rows = 32;
  a.rows_per_chunk = (int)rows;
  int chunks = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_dw_segment& s = p->seg[i];
    DwgSegDev& d = a.seg[i];
    d.x = (const uint16_t*)s.x; d.dy = (const uint16_t*)s.y;
    d.N = s.N; d.H = s.H; d.W = s.W; d.Ho = s.Ho; d.Wo = s.Wo; d.P = s.N * s.Ho * ((s.Wo + 3) / 4);
    d.chunk_begin = chunks;
    chunks += (int)rn_cdiv(d.P, rows);
  }
  a.total_chunks = chunks;
  return 0;
}

extern "C" size_t rn_depthwise_wgrad_workspace_bytes(const rn_dw_problem* p) {
  DwgArgs a;
  if (dwg_plan(p, a)) return 0;
  return (size_t)a.total_chunks * p->k * p->k * a.C * sizeof(float);
}

// problem: seg.x = layer input, seg.y = gradient wrt the layer output (read only), seg.w/scale/shift unused
extern "C" int rn_depthwise_conv2d_nhwc_wgrad(const rn_dw_problem* p, float* dw, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  DwgArgs a;
  RN_CHECK_ARG(dwg_plan(p, a) == 0 && dw, "rn_depthwise_conv2d_nhwc_wgrad: bad problem (k in {1,3,5}, C %% 8 == 0)");
  const size_t need = rn_depthwise_wgrad_workspace_bytes(p);
  if (!workspace || workspace_bytes < need) {
    rn_set_error("rn_depthwise_conv2d_nhwc_wgrad: workspace %zu < %zu", workspace_bytes, need);
    return RN_ENOMEM;
  }
  a.partial = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  const int cg = a.C / (p->k == 5 ? 4 : 8);
  const dim3 g8(a.total_chunks, (unsigned)rn_cdiv(cg, a.slab_groups)), g4 = g8, blk(256);
  switch (p->k * 10 + p->stride) {
    case 11: hipLaunchKernelGGL((depthwise_wgrad_kernel<1, 8, 1>), g8, blk, 0, st, a); break;
    case 12: hipLaunchKernelGGL((depthwise_wgrad_kernel<1, 8, 2>), g8, blk, 0, st, a); break;
    case 31: hipLaunchKernelGGL((depthwise_wgrad_kernel<3, 8, 1>), g8, blk, 0, st, a); break;
    case 32: hipLaunchKernelGGL((depthwise_wgrad_kernel<3, 8, 2>), g8, blk, 0, st, a); break;
    case 51: hipLaunchKernelGGL((depthwise_wgrad_kernel<5, 4, 1>), g4, blk, 0, st, a); break;
    default: hipLaunchKernelGGL((depthwise_wgrad_kernel<5, 4, 2>), g4, blk, 0, st, a); break;
  }
  RN_CHECK_LAUNCH();
  const long long n = (long long)p->k * p->k * a.C;
  hipLaunchKernelGGL(dwg_reduce_kernel, dim3((unsigned)rn_cdiv(n, 16)), dim3(256), 0, st, (const float*)workspace, n,
                     a.total_chunks, dw);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- squeeze-and-excitation --------------------------------------------------------------------
// pooled[n][c] = bf16(mean over H*W)  (tf.reduce_mean on a bf16 tensor yields bf16)
// grid (channel slabs, N, HW chunks): partial[chunk][n][c] = sum over the chunk's pixels (deterministic);
// se_mid_fwd_kernel adds the chunks in order, divides and rounds.
// A workgroup = SG channel groups (16 bytes each) x RL = 256 / SG pixel lanes of one slab of a chunk's pixels; SG from
// se_slab_plan (8 = 64-channel slabs where they keep the lanes busy; the whole pixel for 40 / 144 / 288 ... channels).
__global__ void __launch_bounds__(256)
se_pool_kernel(const uint4* __restrict__ x, int HW, int C8, int rows_per_chunk, int SG, float* __restrict__ partial) {
  const int n = blockIdx.y, slab = blockIdx.x;
  const int RL = 256 / SG;
  const int rl = threadIdx.x / SG, cg = threadIdx.x - rl * SG;
  const int c8 = slab * SG + cg;
  const int p0 = blockIdx.z * rows_per_chunk;
  const int p1 = p0 + rows_per_chunk < HW ? p0 + rows_per_chunk : HW;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
  if (c8 < C8 && rl < RL)
    for (int p = p0 + rl; p < p1; p += RL) {
      const bf8 v = unpack8(x[((long long)n * HW + p) * C8 + c8]);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += v.v[q];
    }
  __shared__ float red[256 * 8 + 256];   // [RL][SG * 8 + 1]
  const int W = SG * 8 + 1;
  if (rl < RL) {
#pragma unroll
    for (int q = 0; q < 8; ++q) red[rl * W + cg * 8 + q] = acc[q];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < SG * 8; c += 256) {
    float t = 0.0f;
    for (int r = 0; r < RL; ++r) t += red[r * W + c];
    const int ch = slab * SG * 8 + c;
    if (ch < C8 * 8) partial[((long long)blockIdx.z * gridDim.y + n) * C8 * 8 + ch] = t;
  }
}
// channel groups per slab / slabs per pixel for the two pooling kernels (the policy of rn_train.hip::bn_slab_plan)
static void se_slab_plan(int C8, long long chunk_groups, int* slab_groups, int* nslab) {
  const int old_n = (C8 + 7) / 8;
  auto score = [&](int n) {
    const int g = (C8 + n - 1) / n;
    return (double)C8 * (256 / g) / ((double)n * 256);
  };
  *slab_groups = C8 < 8 ? C8 : 8;
  *nslab = old_n;
  if (C8 >= 8 && (double)C8 / (8.0 * old_n) >= 0.95) return;
  double best = 0.0;
  for (int n = 1; n <= old_n; ++n) {
    const int g = (C8 + n - 1) / n;
    if (g > 64 || (g < 8 && n > 1)) continue;
    if (score(n) > best) best = score(n);
  }
  for (int n = 1; n <= old_n; ++n) {
    const int g = (C8 + n - 1) / n;
    if (g > 64 || (g < 8 && n > 1) || score(n) < best - 0.01) continue;
    if (chunk_groups * n >= 512 || C8 < 8) {
      *slab_groups = g;
      *nslab = n;
      return;
    }
  }
}

static int se_chunks(int HW, int* rows_per_chunk) {
  int chunks = (HW + 1023) / 1024;            // >= 1024 pixels per chunk
  if (chunks > 32) chunks = 32;
  if (chunks < 1) chunks = 1;
  int rows = (HW + chunks - 1) / chunks;
  rows = (rows + 31) / 32 * 32;
  *rows_per_chunk = rows;
  return (HW + rows - 1) / rows;
}

// pooled[n][c] = bf16(sum of the chunk partials / HW), gate[n][c] = bf16(sigmoid(W2 . bf16(swish(W1 . pooled + b1)) + b2)):
// ONE launch, one workgroup of 1 024 threads per image — per-image work of a few hundred KFLOP.  pooled -> LDS (and the
// state buffer, for the backward pass); fc1 with a wavefront per reduced channel j: h1 = bf16(W1[j] . pooled + b1[j]),
// a = bf16(swish(h1)); fc2 with a thread per channel.  (Until round 6 three launches — chunk sums, fc1, fc2: 5 - 7 us each
// plus two gaps, in a block that runs 26 times per EfficientNet-B3 pass; the very first version, one workgroup of 256
// threads per image doing everything with a thread per output, took up to 0.4 ms.)  C + se floats of LDS: the callers
// check the 64 KB.
#define SE_MID_THREADS 1024
// grid (N, S): the S workgroups of an image each compute pooled and fc1 in full (cheap, coalesced; a grid of N workgroups
// alone left most of the chip idle behind the serial j-loop of fc2) and fc2 for their own slice of `cps` channels
__global__ void __launch_bounds__(SE_MID_THREADS)
se_mid_fwd_kernel(const float* __restrict__ partial, int chunks, float scale, const uint16_t* __restrict__ w1,
                  const float* __restrict__ b1, const uint16_t* __restrict__ w2, const float* __restrict__ b2, int N, int C,
                  int se, int cps, float* __restrict__ pooled, float* __restrict__ gate, float* __restrict__ h1,
                  float* __restrict__ av) {
  extern __shared__ float se_lds[];   // pooled[C] | a[se]
  float* s_p = se_lds;
  float* s_a = se_lds + C;
  const int n = blockIdx.x, split = blockIdx.y;
  const long long nel = (long long)N * C;
  for (int c = threadIdx.x; c < C; c += SE_MID_THREADS) {
    float t = 0.0f;
    for (int k = 0; k < chunks; ++k) t += partial[(long long)k * nel + (long long)n * C + c];
    t *= scale;
    t = rn_bf16_to_f32(rn_f32_to_bf16(t));
    s_p[c] = t;
    if (split == 0) pooled[(long long)n * C + c] = t;
  }
  __syncthreads();
  // fc1: a wavefront per reduced channel j, 16-byte loads of W1[j] (8 weights per lane and load, all loads independent)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < se; j += SE_MID_THREADS / 64) {
    const uint4* w = (const uint4*)(w1 + (long long)j * C);
    float acc = 0.0f;
    for (int c8 = lane; c8 < (C >> 3); c8 += 64) {
      const bf8 wv = unpack8(w[c8]);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += s_p[c8 * 8 + q] * wv.v[q];
    }
    acc = rn_wave_sum(acc);
    if (lane == 0) {
      float v = rn_bf16_to_f32(rn_f32_to_bf16(acc + b1[j]));
      if (split == 0) h1[(long long)n * se + j] = v;
      v = v / (1.0f + __expf(-v));
      v = rn_bf16_to_f32(rn_f32_to_bf16(v));
      if (split == 0) av[(long long)n * se + j] = v;
      s_a[j] = v;
    }
  }
  __syncthreads();
  // fc2: a wavefront per channel of the slice — W2[c] is one contiguous row of se weights, read by the lanes side by side
  // (a thread per channel walked its row with se dependent-latency 2-byte loads: 10 - 80 us per launch)
  const int c1 = (split + 1) * cps < C ? (split + 1) * cps : C;
  for (int c = split * cps + wave; c < c1; c += SE_MID_THREADS / 64) {
    const uint16_t* w = w2 + (long long)c * se;
    float acc = 0.0f;
    for (int j = lane; j < se; j += 64) acc += s_a[j] * rn_bf16_to_f32(w[j]);
    acc = rn_wave_sum(acc);
    if (lane == 0) {
      float v = rn_bf16_to_f32(rn_f32_to_bf16(acc + b2[c]));
      v = 1.0f / (1.0f + __expf(-v));
      gate[(long long)n * C + c] = rn_bf16_to_f32(rn_f32_to_bf16(v));
    }
  }
}
// channels per workgroup of the two per-image kernels: enough workgroups to spread over the chip, slices of >= 64 channels
static int se_mid_cps(int N, int C, int* splits) {
  int S = (255 + N) / N;
  if (S > (C + 63) / 64) S = (C + 63) / 64;
  if (S < 1) S = 1;
  int cps = ((C + S - 1) / S + 63) / 64 * 64;
  *splits = (C + cps - 1) / cps;
  return cps;
}

__global__ void __launch_bounds__(DW_THREADS)
se_gate_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, const float* __restrict__ gate, int HW, int C8,
               long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const long long n = i / ((long long)HW * C8);
    bf8 v = unpack8(x[i]);
    const float* g = gate + (n * C8 + c) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) v.v[q] *= g[q];
    y[i] = pack8(v);
  }
}

// workspace / state layout: pooled[N][C], gate[N][C], h1[N][se], a[N][se] (se <= C), all f32
// + 32 chunk partials [32][N][C] behind the four [N][C]-sized arrays
extern "C" size_t rn_se_workspace_bytes(int N, int C) { return (size_t)N * C * (4 + 32) * sizeof(float); }

static int se_forward(const void* x, void* y, int N, int HW, int C, const void* w_reduce, const float* b_reduce,
                      const void* w_expand, const float* b_expand, int se, float* state, hipStream_t st) {
  float* pooled = state;
  float* gate = pooled + (size_t)N * C;
  float* h1 = gate + (size_t)N * C;
  float* av = h1 + (size_t)N * se;
  {
    int rows = 0;
    const int chunks = se_chunks(HW, &rows);
    float* partial = state + (size_t)N * C * 4;
    int sg = 8, nslab = 1;
    se_slab_plan(C / 8, (long long)N * chunks, &sg, &nslab);
    hipLaunchKernelGGL(se_pool_kernel, dim3(nslab, N, chunks), dim3(256), 0, st, (const uint4*)x, HW, C / 8, rows, sg,
                       partial);
    RN_CHECK_LAUNCH();
    int splits = 1;
    const int cps = se_mid_cps(N, C, &splits);
    hipLaunchKernelGGL(se_mid_fwd_kernel, dim3(N, splits), dim3(SE_MID_THREADS), (size_t)(C + se) * sizeof(float), st, partial,
                       chunks, 1.0f / (float)HW, (const uint16_t*)w_reduce, b_reduce, (const uint16_t*)w_expand, b_expand, N, C,
                       se, cps, pooled, gate, h1, av);
    RN_CHECK_LAUNCH();
  }
  const long long total = (long long)N * HW * (C / 8);
  long long blocks = rn_cdiv(total, DW_THREADS);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)blocks), dim3(DW_THREADS), 0, st, (const uint4*)x, (uint4*)y, gate,
                     HW, C / 8, total);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_squeeze_excite_inplace(void* x, int N, int HW, int C, const void* w_reduce, const float* b_reduce,
                                         const void* w_expand, const float* b_expand, int se, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(x && w_reduce && b_reduce && w_expand && b_expand && N > 0 && HW > 0 && C % 8 == 0 && se > 0 && se <= C,
               "rn_squeeze_excite_inplace: bad argument");
  if (!workspace || workspace_bytes < rn_se_workspace_bytes(N, C)) {
    rn_set_error("rn_squeeze_excite_inplace: workspace too small");
    return RN_ENOMEM;
  }
  RN_CHECK_ARG((size_t)(C + se) * 4 <= 64 * 1024, "rn_squeeze_excite_inplace: C=%d too large", C);
  return se_forward(x, x, N, HW, C, w_reduce, b_reduce, w_expand, b_expand, se, (float*)workspace, (hipStream_t)stream);
}

// training forward: out of place (the backward needs the un-gated input), `state` (rn_se_workspace_bytes)
// keeps pooled / gate / h1 / a for rn_squeeze_excite_bwd
extern "C" int rn_squeeze_excite_fwd(const void* x, void* y, int N, int HW, int C, const void* w_reduce,
                                     const float* b_reduce, const void* w_expand, const float* b_expand, int se,
                                     void* state, size_t state_bytes, void* stream) {
  RN_CHECK_ARG(x && y && w_reduce && b_reduce && w_expand && b_expand && N > 0 && HW > 0 && C % 8 == 0 && se > 0 &&
                   se <= C && (size_t)(C + se) * 4 <= 64 * 1024, "rn_squeeze_excite_fwd: bad argument");
  if (!state || state_bytes < rn_se_workspace_bytes(N, C)) {
    rn_set_error("rn_squeeze_excite_fwd: state buffer too small");
    return RN_ENOMEM;
  }
  return se_forward(x, y, N, HW, C, w_reduce, b_reduce, w_expand, b_expand, se, (float*)state, (hipStream_t)stream);
}

// ---- squeeze-and-excitation backward ------------------------------------------------------------------
// y = x * g[n][c], g = sigmoid(h2), h2 = W2 a + b2, a = swish(h1), h1 = W1 p + b1, p = mean_hw x
// dgate[n][c] = sum_hw dy * x
__global__ void __launch_bounds__(256)
se_bwd_pool_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy, int HW, int C8, int rows_per_chunk, int SG,
                   float* __restrict__ partial) {
  const int n = blockIdx.y, slab = blockIdx.x;   // (slab geometry: se_pool_kernel)
  const int RL = 256 / SG;
  const int rl = threadIdx.x / SG, cg = threadIdx.x - rl * SG;
  const int c8 = slab * SG + cg;
  const int p0 = blockIdx.z * rows_per_chunk;
  const int p1 = p0 + rows_per_chunk < HW ? p0 + rows_per_chunk : HW;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
  if (c8 < C8 && rl < RL)
    for (int p = p0 + rl; p < p1; p += RL) {
      const long long o = ((long long)n * HW + p) * C8 + c8;
      const bf8 v = unpack8(x[o]), g = unpack8(dy[o]);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += v.v[q] * g.v[q];
    }
  __shared__ float red[256 * 8 + 256];   // [RL][SG * 8 + 1]
  const int W = SG * 8 + 1;
  if (rl < RL) {
#pragma unroll
    for (int q = 0; q < 8; ++q) red[rl * W + cg * 8 + q] = acc[q];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < SG * 8; c += 256) {
    float t = 0.0f;
    for (int r = 0; r < RL; ++r) t += red[r * W + c];
    const int ch = slab * SG * 8 + c;
    if (ch < C8 * 8) partial[((long long)blockIdx.z * gridDim.y + n) * C8 * 8 + ch] = t;
  }
}

// dgate[n][c] = sum of the chunk partials; dh2 = dgate * g * (1 - g) (thread per channel); dh1[n][j] = (W2^T dh2)[j] *
// swish'(h1) (wavefront per reduced channel); dp[n][c] = (W1^T dh1)[c] (thread per channel): one launch, a workgroup of
// 1 024 threads per image (see se_mid_fwd_kernel; four launches until round 6)
__global__ void __launch_bounds__(SE_MID_THREADS)
se_mid_bwd_kernel(const float* __restrict__ partial, int chunks, const float* __restrict__ state, const uint16_t* __restrict__ w1,
                  const uint16_t* __restrict__ w2, int N, int C, int se, int cps, float* __restrict__ dh2,
                  float* __restrict__ dh1, float* __restrict__ dp) {
  extern __shared__ float se_lds[];   // dh2[C] | dh1[se] | shares[16][se]
  float* s_d2 = se_lds;
  float* s_d1 = se_lds + C;
  const int n = blockIdx.x, split = blockIdx.y;   // (N, S) as in se_mid_fwd_kernel: dh2 / dh1 in full, dp for a slice
  const long long nel = (long long)N * C;
  for (int c = threadIdx.x; c < C; c += SE_MID_THREADS) {
    float t = 0.0f;
    for (int k = 0; k < chunks; ++k) t += partial[(long long)k * nel + (long long)n * C + c];
    const float g = state[nel + (long long)n * C + c];
    const float d = t * g * (1.0f - g);
    s_d2[c] = d;
    if (split == 0) dh2[(long long)n * C + c] = d;
  }
  __syncthreads();
  // dh1 = W2^T dh2: wavefront w walks the rows c = w, w + 16, ... of W2 (contiguous, the lanes side by side over j) and keeps
  // its share of the sum for the lane's j; the 16 shares are added in wavefront order (deterministic)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* s_part = s_d1 + se;   // [16][se]
  for (int j0 = 0; j0 < se; j0 += 64) {
    const int j = j0 + lane;
    float acc = 0.0f;
    if (j < se)
      for (int c = wave; c < C; c += SE_MID_THREADS / 64) acc += s_d2[c] * rn_bf16_to_f32(w2[(long long)c * se + j]);
    if (j < se) s_part[wave * se + j] = acc;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < se; j += SE_MID_THREADS) {
    float acc = 0.0f;
    for (int w = 0; w < SE_MID_THREADS / 64; ++w) acc += s_part[w * se + j];
    const float u = state[2 * nel + (long long)n * se + j];   // h1
    const float sg = 1.0f / (1.0f + __expf(-u));
    const float d = acc * (sg + u * sg * (1.0f - sg));
    if (split == 0) dh1[(long long)n * se + j] = d;
    s_d1[j] = d;
  }
  __syncthreads();
  // dp = W1^T dh1 for the slice: a thread per pair of channels, W1[j] read side by side (4 bytes per lane), 8 rows in flight
  const int c1 = (split + 1) * cps < C ? (split + 1) * cps : C;
  for (int c = split * cps + 2 * threadIdx.x; c < c1; c += 2 * SE_MID_THREADS) {
    float a0 = 0.0f, a1 = 0.0f;
    int j = 0;
    for (; j + 8 <= se; j += 8) {
      unsigned u[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = *(const unsigned*)(w1 + (long long)(j + k) * C + c);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a0 += s_d1[j + k] * rn_bf16_to_f32((uint16_t)(u[k] & 0xffffu));
        a1 += s_d1[j + k] * rn_bf16_to_f32((uint16_t)(u[k] >> 16));
      }
    }
    for (; j < se; ++j) {
      const unsigned u = *(const unsigned*)(w1 + (long long)j * C + c);
      a0 += s_d1[j] * rn_bf16_to_f32((uint16_t)(u & 0xffffu));
      a1 += s_d1[j] * rn_bf16_to_f32((uint16_t)(u >> 16));
    }
    dp[(long long)n * C + c] = a0;
    dp[(long long)n * C + c + 1] = a1;
  }
}

// parameter gradients: sums over the N images (thread per element, fixed order)
__global__ void __launch_bounds__(256)
se_bwd_wgrad_kernel(const float* __restrict__ state, const float* __restrict__ dh2, const float* __restrict__ dh1, int N,
                    int C, int se, float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                    float* __restrict__ db2) {
  const float* pooled = state;
  const float* av = state + (size_t)2 * N * C + (size_t)N * se;
  const long long n1 = (long long)se * C;
  const long long total = 2 * n1 + se + C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float t = 0.0f;
    if (i < n1) {                       // dw1[j][c]
      const int j = (int)(i / C), c = (int)(i - (long long)j * C);
      for (int n = 0; n < N; ++n) t += dh1[(long long)n * se + j] * pooled[(long long)n * C + c];
      dw1[i] = t;
    } else if (i < 2 * n1) {            // dw2[c][j]
      const long long k = i - n1;
      const int c = (int)(k / se), j = (int)(k - (long long)c * se);
      for (int n = 0; n < N; ++n) t += dh2[(long long)n * C + c] * av[(long long)n * se + j];
      dw2[k] = t;
    } else if (i < 2 * n1 + se) {
      const int j = (int)(i - 2 * n1);
      for (int n = 0; n < N; ++n) t += dh1[(long long)n * se + j];
      db1[j] = t;
    } else {
      const int c = (int)(i - 2 * n1 - se);
      for (int n = 0; n < N; ++n) t += dh2[(long long)n * C + c];
      db2[c] = t;
    }
  }
}

// dx = dy * g + dp / HW
__global__ void __launch_bounds__(DW_THREADS)
se_bwd_apply_kernel(const uint4* __restrict__ dy, uint4* __restrict__ dx, const float* __restrict__ gate,
                    const float* __restrict__ dp, int HW, int C8, long long total) {
  const float inv = 1.0f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const long long n = i / ((long long)HW * C8);
    bf8 v = unpack8(dy[i]);
    const float* g = gate + (n * C8 + c) * 8;
    const float* d = dp + (n * C8 + c) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) v.v[q] = v.v[q] * g[q] + d[q] * inv;
    dx[i] = pack8(v);
  }
}

// x: the un-gated input of the forward, dy: gradient wrt the gated output, state: from rn_squeeze_excite_fwd.
// dx may alias dy.  dw1 [se][C], db1 [se], dw2 [C][se], db2 [C] are overwritten.  workspace: rn_se_workspace_bytes.
extern "C" int rn_squeeze_excite_bwd(const void* x, const void* dy, void* dx, int N, int HW, int C, const void* w_reduce,
                                     const void* w_expand, int se, const void* state, float* dw1, float* db1,
                                     float* dw2, float* db2, void* workspace, size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(x && dy && dx && w_reduce && w_expand && state && dw1 && db1 && dw2 && db2 && N > 0 && HW > 0 &&
                   C % 8 == 0 && se > 0 && se <= C && (size_t)(C + 17 * se) * 4 <= 64 * 1024,
               "rn_squeeze_excite_bwd: bad argument");
  if (!workspace || workspace_bytes < rn_se_workspace_bytes(N, C)) {
    rn_set_error("rn_squeeze_excite_bwd: workspace too small");
    return RN_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream;
  float* dh2 = (float*)workspace + (size_t)N * C;   // (the first [N][C] slot held dgate until the per-image steps were one launch)
  float* dp = dh2 + (size_t)N * C;
  float* dh1 = dp + (size_t)N * C;
  const float* stf = (const float*)state;
  {
    int rows = 0;
    const int chunks = se_chunks(HW, &rows);
    float* partial = (float*)workspace + (size_t)N * C * 4;
    int sg = 8, nslab = 1;
    se_slab_plan(C / 8, (long long)N * chunks, &sg, &nslab);
    hipLaunchKernelGGL(se_bwd_pool_kernel, dim3(nslab, N, chunks), dim3(256), 0, st, (const uint4*)x,
                       (const uint4*)dy, HW, C / 8, rows, sg, partial);
    RN_CHECK_LAUNCH();
    int splits = 1;
    const int cps = se_mid_cps(N, C, &splits);
    hipLaunchKernelGGL(se_mid_bwd_kernel, dim3(N, splits), dim3(SE_MID_THREADS), (size_t)(C + 17 * se) * sizeof(float), st, partial,
                       chunks, stf, (const uint16_t*)w_reduce, (const uint16_t*)w_expand, N, C, se, cps, dh2, dh1, dp);
    RN_CHECK_LAUNCH();
  }
  const long long nel = 2ll * se * C + se + C;
  hipLaunchKernelGGL(se_bwd_wgrad_kernel, dim3((unsigned)rn_cdiv(nel, 256)), dim3(256), 0, st, stf, dh2, dh1, N, C, se,
                     dw1, db1, dw2, db2);
  RN_CHECK_LAUNCH();
  const long long total = (long long)N * HW * (C / 8);
  long long blocks = rn_cdiv(total, DW_THREADS);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(se_bwd_apply_kernel, dim3((unsigned)blocks), dim3(DW_THREADS), 0, st, (const uint4*)dy, (uint4*)dx,
                     stf + (size_t)N * C, dp, HW, C / 8, total);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
