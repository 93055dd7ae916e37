// rn_depthwise.hip — a18: the EfficientNet / SeparableConv2D building blocks that are not GEMMs.
//   * depthwise k x k convolution (tf.keras.layers.DepthwiseConv2D in MBConvBlock,
//     retinanet/model/backbone/efficientnet.py:372-380, and the depthwise half of SeparableConv2D
//     in fpn_base.py:28-39 / detection_head.py:37-50), TF SAME padding, fused BN scale/shift + act;
//   * squeeze-and-excitation (efficientnet.py:222-265): global average pool, the two 1x1 "convs" on
//     the pooled vector (swish, sigmoid) and the per-(image, channel) gate.
// All HBM-bound: bf16 NHWC, 8 channels (16 B) per thread, fp32 math.  Grouped like the GEMM convs so
// the five pyramid levels of a shared separable head conv go out in one launch.
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
__device__ __forceinline__ float act_exact(float v, int act) {
  switch (act) {
    case RN_ACT_RELU: return fmaxf(v, 0.0f);
    case RN_ACT_RELU6: return fminf(fmaxf(v, 0.0f), 6.0f);
    case RN_ACT_SWISH: return v / (1.0f + __expf(-v));
    default: return v;
  }
}

struct DwSegDev {
  const uint4* x; const uint4* w; uint4* y; const float* scale; const float* shift;
  int N, H, W, C8, Ho, Wo;
  long long begin;
};
struct DwArgs {
  int k, stride, pt, pl, act, nseg;
  long long total;
  DwSegDev seg[RN_CONV_MAX_SEGMENTS];
};

__global__ void __launch_bounds__(DW_THREADS) depthwise_kernel(const DwArgs a) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < a.total;
       i += (long long)gridDim.x * blockDim.x) {
    int si = 0;
    while (si + 1 < a.nseg && i >= a.seg[si + 1].begin) ++si;
    const DwSegDev& s = a.seg[si];
    long long t = i - s.begin;
    const int c = (int)(t % s.C8);
    t /= s.C8;
    const int ox = (int)(t % s.Wo);
    t /= s.Wo;
    const int oy = (int)(t % s.Ho);
    const int n = (int)(t / s.Ho);
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
    for (int r = 0; r < a.k; ++r) {
      const int iy = oy * a.stride - a.pt + r;
      if ((unsigned)iy >= (unsigned)s.H) continue;
      for (int ss = 0; ss < a.k; ++ss) {
        const int ix = ox * a.stride - a.pl + ss;
        if ((unsigned)ix >= (unsigned)s.W) continue;
        const bf8 xv = unpack8(s.x[(((long long)n * s.H + iy) * s.W + ix) * s.C8 + c]);
        const bf8 wv = unpack8(s.w[(long long)(r * a.k + ss) * s.C8 + c]);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += xv.v[q] * wv.v[q];
      }
    }
    bf8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = acc[q];
      if (s.scale) v *= s.scale[c * 8 + q];
      if (s.shift) v += s.shift[c * 8 + q];
      o.v[q] = act_exact(v, a.act);
    }
    s.y[(((long long)n * s.Ho + oy) * s.Wo + ox) * s.C8 + c] = pack8(o);
  }
}

extern "C" int rn_depthwise_conv2d_nhwc_fwd(const rn_dw_problem* p, void* stream) {
  RN_CHECK_ARG(p && p->num_segments >= 1 && p->num_segments <= RN_CONV_MAX_SEGMENTS && p->k >= 1 && p->stride >= 1,
               "rn_depthwise_conv2d_nhwc_fwd: bad problem");
  DwArgs a;
  a.k = p->k; a.stride = p->stride; a.pt = p->pad_top; a.pl = p->pad_left; a.act = p->act; a.nseg = p->num_segments;
  long long off = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_dw_segment& s = p->seg[i];
    RN_CHECK_ARG(s.x && s.w && s.y && s.C % 8 == 0 && s.N > 0 && s.H > 0 && s.W > 0 && s.Ho > 0 && s.Wo > 0,
                 "rn_depthwise_conv2d_nhwc_fwd: segment %d bad (C %% 8 == 0)", i);
    DwSegDev& d = a.seg[i];
    d.x = (const uint4*)s.x; d.w = (const uint4*)s.w; d.y = (uint4*)s.y; d.scale = s.scale; d.shift = s.shift;
    d.N = s.N; d.H = s.H; d.W = s.W; d.C8 = s.C / 8; d.Ho = s.Ho; d.Wo = s.Wo;
    d.begin = off;
    off += (long long)s.N * s.Ho * s.Wo * (s.C / 8);
  }
  a.total = off;
  long long blocks = rn_cdiv(off, DW_THREADS);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(depthwise_kernel, dim3((unsigned)blocks), dim3(DW_THREADS), 0, (hipStream_t)stream, a);
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

// ---- squeeze-and-excitation --------------------------------------------------------------------
// pooled[n][c] = bf16(mean over H*W)  (tf.reduce_mean on a bf16 tensor yields bf16)
__global__ void __launch_bounds__(256)
se_pool_kernel(const uint4* __restrict__ x, int HW, int C8, float* __restrict__ pooled) {
  const int n = blockIdx.y, slab = blockIdx.x;  // 8 channel groups (64 channels) per block
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c8 = slab * 8 + cg;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.0f;
  if (c8 < C8)
    for (int p = rl; p < HW; p += 32) {
      const bf8 v = unpack8(x[((long long)n * HW + p) * C8 + c8]);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += v.v[q];
    }
  __shared__ float red[32][65];
#pragma unroll
  for (int q = 0; q < 8; ++q) red[rl][cg * 8 + q] = acc[q];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.0f;
    for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
    const int ch = slab * 64 + threadIdx.x;
    if (ch < C8 * 8) pooled[(long long)n * C8 * 8 + ch] = rn_bf16_to_f32(rn_f32_to_bf16(t / (float)HW));
  }
}

// gate[n][c] = bf16(sigmoid(W2 . bf16(swish(W1 . pooled + b1)) + b2)); one workgroup per image
__global__ void __launch_bounds__(256)
se_fc_kernel(const float* __restrict__ pooled, const uint16_t* __restrict__ w1 /*[se][C]*/,
             const float* __restrict__ b1, const uint16_t* __restrict__ w2 /*[C][se]*/,
             const float* __restrict__ b2, int C, int se, float* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sp = (float*)smem;  // [C]
  float* sh = sp + C;        // [se]
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sp[c] = pooled[(long long)n * C + c];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < se; j += blockDim.x / 64) {
    float acc = 0.0f;
    for (int c = lane; c < C; c += 64) acc += sp[c] * rn_bf16_to_f32(w1[(long long)j * C + c]);
    acc = rn_wave_sum(acc);
    if (lane == 0) {
      float v = acc + b1[j];
      v = rn_bf16_to_f32(rn_f32_to_bf16(v));
      v = v / (1.0f + __expf(-v));
      sh[j] = rn_bf16_to_f32(rn_f32_to_bf16(v));
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.0f;
    for (int j = 0; j < se; ++j) acc += sh[j] * rn_bf16_to_f32(w2[(long long)c * se + j]);
    float v = acc + b2[c];
    v = rn_bf16_to_f32(rn_f32_to_bf16(v));
    v = 1.0f / (1.0f + __expf(-v));
    gate[(long long)n * C + c] = rn_bf16_to_f32(rn_f32_to_bf16(v));
  }
}

__global__ void __launch_bounds__(DW_THREADS)
se_gate_kernel(uint4* __restrict__ x, const float* __restrict__ gate, int HW, int C8, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const long long n = i / ((long long)HW * C8);
    bf8 v = unpack8(x[i]);
    const float* g = gate + (n * C8 + c) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) v.v[q] *= g[q];
    x[i] = pack8(v);
  }
}

extern "C" size_t rn_se_workspace_bytes(int N, int C) { return (size_t)N * C * 2 * sizeof(float); }

extern "C" int rn_squeeze_excite_inplace(void* x, int N, int HW, int C, const void* w_reduce, const float* b_reduce,
                                         const void* w_expand, const float* b_expand, int se, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(x && w_reduce && b_reduce && w_expand && b_expand && N > 0 && HW > 0 && C % 8 == 0 && se > 0,
               "rn_squeeze_excite_inplace: bad argument");
  if (!workspace || workspace_bytes < rn_se_workspace_bytes(N, C)) {
    rn_set_error("rn_squeeze_excite_inplace: workspace too small");
    return RN_ENOMEM;
  }
  RN_CHECK_ARG((size_t)(C + se) * 4 <= 64 * 1024, "rn_squeeze_excite_inplace: C=%d too large", C);
  hipStream_t st = (hipStream_t)stream;
  float* pooled = (float*)workspace;
  float* gate = pooled + (size_t)N * C;
  hipLaunchKernelGGL(se_pool_kernel, dim3((C + 63) / 64, N), dim3(256), 0, st, (const uint4*)x, HW, C / 8, pooled);
  RN_CHECK_LAUNCH();
  hipLaunchKernelGGL(se_fc_kernel, dim3(N), dim3(256), (size_t)(C + se) * 4, st, pooled, (const uint16_t*)w_reduce,
                     b_reduce, (const uint16_t*)w_expand, b_expand, C, se, gate);
  RN_CHECK_LAUNCH();
  const long long total = (long long)N * HW * (C / 8);
  long long blocks = rn_cdiv(total, DW_THREADS);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)blocks), dim3(DW_THREADS), 0, st, (uint4*)x, gate, HW, C / 8,
                     total);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
