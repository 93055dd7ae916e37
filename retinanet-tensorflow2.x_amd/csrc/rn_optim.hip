// rn_optim.hip — a10: the optimizer side of Executor._train_step (retinanet/executor.py:409-441)
// as multi-tensor kernels over ONE flat fp32 parameter arena (the reference walks 295 separate
// tf.Variables; here params / grads / momentum / EMA are four flat buffers and a tensor is a
// (offset, size) segment — one launch per stage instead of ~300).
//   stage 1  g <- g + (alpha/R) * w for weight-decay tensors (l2_loss gradient, executor.py:296-299,
//            417-421 with the 1/R of per_replica_loss), per-block sum of squares;
//   stage 2  per-tensor norm -> clip_by_norm factor, global norm of the clipped tensors ->
//            clip_by_global_norm factor (executor.py:401-407; tf semantics c/max(norm,c));
//   stage 3  g <- g * factor[tensor]                      (then the data-parallel all-reduce SUM)
//   stage 4  Keras SGD momentum  v <- m v - lr g ; w <- w + v   (optimizers/builder.py:45)
//            tfa MovingAverage   ema <- ema - (1-d)(ema - w)    (optimizers/builder.py:51-54)
//            and the bf16 compute copy of w for the next forward.
// Deterministic: block partials are added in index order.  HBM-bound: 34.4 M params.
#include "rn_common.h"

#define OPT_THREADS 256
#define OPT_CHUNK 8192  // elements per block

struct OptSeg { long long offset, size; int wd, block_begin, nblocks, pad_; long long bf16_offset; };

// partial[b] = sum of squares of the block's (unscaled, weight-decayed) gradients; partial_w[b] = sum of squares of
// its weights when the tensor is weight-decayed (the l2-regularization term the reference logs, executor.py:296-299)
__global__ void __launch_bounds__(OPT_THREADS)
optim_sqnorm_kernel(float* __restrict__ g, const float* __restrict__ w, const OptSeg* __restrict__ segs,
                    const int* __restrict__ block_seg, int block0, float wdc, float unscale,
                    double* __restrict__ partial, double* __restrict__ partial_w, float* __restrict__ local_copy) {
  const int blk = block0 + blockIdx.x;
  const int si = block_seg[blk];
  const OptSeg s = segs[si];
  const long long b0 = (long long)(blk - s.block_begin) * OPT_CHUNK;
  long long b1 = b0 + OPT_CHUNK;
  if (b1 > s.size) b1 = s.size;
  float acc = 0.0f, accw = 0.0f;
  const bool touch = s.wd || unscale != 1.0f;
  // LossScaleOptimizer.get_unscaled_gradients (executor.py:429-430), then the l2 gradient of the decayed tensors
#define SQNORM_ELEM(gv_, wv_)                 \
  do {                                        \
    float v__ = (gv_) * unscale;              \
    if (s.wd) {                               \
      v__ += wdc * (wv_);                     \
      accw += (wv_) * (wv_);                  \
    }                                         \
    (gv_) = v__;                              \
    acc += v__ * v__;                         \
  } while (0)
  if (((s.offset + b0) & 3) == 0) {   // 16 bytes per lane (the arena aligns every tensor to 4 elements); scalar tail
    const long long base = s.offset + b0;
    const int n = (int)(b1 - b0), n4 = n >> 2;
    float4* g4 = (float4*)(g + base);
    const float4* w4 = (const float4*)(w + base);
    float4* l4 = local_copy ? (float4*)(local_copy + base) : nullptr;   // this rank's gradient, kept for the clip correction
    for (int i = threadIdx.x; i < n4; i += OPT_THREADS) {
      float4 gv = g4[i], wv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (s.wd) wv = w4[i];
      SQNORM_ELEM(gv.x, wv.x); SQNORM_ELEM(gv.y, wv.y); SQNORM_ELEM(gv.z, wv.z); SQNORM_ELEM(gv.w, wv.w);
      if (touch) g4[i] = gv;
      if (l4) l4[i] = gv;
    }
    if ((int)threadIdx.x < (n & 3)) {
      const long long o = base + 4 * n4 + threadIdx.x;
      float gv = g[o];
      const float wv = s.wd ? w[o] : 0.0f;
      SQNORM_ELEM(gv, wv);
      if (touch) g[o] = gv;
      if (local_copy) local_copy[o] = gv;
    }
  } else {
    for (long long i = b0 + threadIdx.x; i < b1; i += OPT_THREADS) {
      float gv = g[s.offset + i];
      const float wv = s.wd ? w[s.offset + i] : 0.0f;
      SQNORM_ELEM(gv, wv);
      if (touch) g[s.offset + i] = gv;
      if (local_copy) local_copy[s.offset + i] = gv;
    }
  }
#undef SQNORM_ELEM
  __shared__ double red[2][OPT_THREADS / 64];
  const double d = rn_wave_sum_d((double)acc), dw = rn_wave_sum_d((double)accw);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d; red[1][threadIdx.x >> 6] = dw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0, tw = 0.0;
    for (int k = 0; k < OPT_THREADS / 64; ++k) { t += red[0][k]; tw += red[1][k]; }
    partial[blk] = t;
    partial_w[blk] = tw;
  }
}

// metrics f32[8]: [0] global norm after clipping, [1] before the global clip, [2] the global factor,
// [3] l2-regularization = alpha * sum over decayed tensors of ||w||^2 / 2, [4] 1 if any per-tensor or the global
// factor != 1 on this rank ("the clip fired"), [5] 1 if the gradient norm is not finite (LossScaleOptimizer skips
// the step).  flags (optional, device f32[2]): [0] = metrics[4] or metrics[5], [1] = metrics[5] — the slots that ride in
// the last gradient bucket of the overlapped all-reduce.
__global__ void __launch_bounds__(256)
optim_factors_kernel(const OptSeg* __restrict__ segs, int nseg, const double* __restrict__ partial,
                     const double* __restrict__ partial_w, float clip, float wd_alpha, float* __restrict__ factor,
                     float* __restrict__ metrics, float* __restrict__ flags) {
  __shared__ double s_sq[256], s_w[256];
  __shared__ int s_fired[256];
  double local = 0.0, localw = 0.0;
  int fired = 0;
  for (int t = threadIdx.x; t < nseg; t += blockDim.x) {
    const OptSeg s = segs[t];
    double sq = 0.0, sw = 0.0;
    // (the adds stay in block order; unrolled so that eight loads are in flight — the thread that owns the largest
    // tensor walks ~300 partials, one memory round trip each before)
#pragma unroll 8
    for (int b = 0; b < s.nblocks; ++b) { sq += partial[s.block_begin + b]; sw += partial_w[s.block_begin + b]; }
    const float norm = (float)sqrt(sq);
    const float f = clip > 0.0f ? clip / fmaxf(norm, clip) : 1.0f;  // tf.clip_by_norm
    factor[t] = f;
    fired |= f != 1.0f;
    const float cn = norm * f;
    local += (double)cn * (double)cn;
    localw += sw;
  }
  s_sq[threadIdx.x] = local;
  s_w[threadIdx.x] = localw;
  s_fired[threadIdx.x] = fired;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0, totw = 0.0;
    int any = 0;
    for (int k = 0; k < 256; ++k) { tot += s_sq[k]; totw += s_w[k]; any |= s_fired[k]; }
    const float gn = (float)sqrt(tot);
    const float F = clip > 0.0f ? clip / fmaxf(gn, clip) : 1.0f;    // tf.clip_by_global_norm
    const bool finite = gn == gn && gn <= 3.0e38f;
    metrics[0] = gn * F;  // global norm of the clipped gradients (executor.py:440 multiplies by R)
    metrics[1] = gn;      // before the global clip
    metrics[2] = F;
    metrics[3] = (float)(0.5 * (double)wd_alpha * totw);
    metrics[4] = (any || F != 1.0f) ? 1.0f : 0.0f;
    metrics[5] = finite ? 0.0f : 1.0f;
    // flags[0] also fires on a non-finite norm (NaN makes no factor != 1: fmaxf drops it), so that "flags[0] == 0 on every
    // rank" means "no correction AND nothing to skip": the predicate of the optimistic SGD launch (train_engine.py)
    if (flags) { flags[0] = (metrics[4] != 0.0f || !finite) ? 1.0f : 0.0f; flags[1] = metrics[5]; }
    s_sq[0] = (double)F;
  }
  __syncthreads();
  const float F = (float)s_sq[0];
  for (int t = threadIdx.x; t < nseg; t += blockDim.x) factor[t] *= F;
}

// g <- g * factor[tensor] in place (out == nullptr), or out <- (factor[tensor] - 1) * g: the correction this rank
// owes an all-reduce that already summed its UNclipped gradients (the optimistic-clip overlap, see rn_optim_clip_*)
__global__ void __launch_bounds__(OPT_THREADS)
optim_scale_kernel(float* __restrict__ g, float* __restrict__ out, const OptSeg* __restrict__ segs,
                   const int* __restrict__ block_seg, const float* __restrict__ factor) {
  const int si = block_seg[blockIdx.x];
  const OptSeg s = segs[si];
  const float f = factor[si];
  if (f == 1.0f && !out) return;
  const long long b0 = (long long)(blockIdx.x - s.block_begin) * OPT_CHUNK;
  long long b1 = b0 + OPT_CHUNK;
  if (b1 > s.size) b1 = s.size;
  if (out) {
    const float c = f - 1.0f;
    for (long long i = b0 + threadIdx.x; i < b1; i += OPT_THREADS) out[s.offset + i] = c * g[s.offset + i];
  } else {
    for (long long i = b0 + threadIdx.x; i < b1; i += OPT_THREADS) g[s.offset + i] *= f;
  }
}

__global__ void __launch_bounds__(OPT_THREADS)
optim_sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, float* __restrict__ ema,
                 uint16_t* __restrict__ wbf16, const OptSeg* __restrict__ segs, const int* __restrict__ block_seg,
                 float lr, float momentum, float ema_decay, int use_ema, int nesterov,
                 const float* __restrict__ skip_flag) {
  // LossScaleOptimizer.apply_gradients: a step whose gradients are not finite on ANY replica is dropped
  if (skip_flag && skip_flag[0] != 0.0f) return;
  const int si = block_seg[blockIdx.x];
  const OptSeg s = segs[si];
  const long long b0 = (long long)(blockIdx.x - s.block_begin) * OPT_CHUNK;
  long long b1 = b0 + OPT_CHUNK;
  if (b1 > s.size) b1 = s.size;
#define SGD_ELEM(w_, g_, v_, e_)                                                                       \
  do {                                                                                                 \
    const float vel__ = momentum * (v_) - lr * (g_);                                                   \
    /* Keras SGD: w += v  (momentum), or w += momentum * v - lr * g with the NEW v (nesterov=True) */   \
    const float nw__ = (w_) + (nesterov ? momentum * vel__ - lr * (g_) : vel__);                       \
    (v_) = vel__;                                                                                      \
    (w_) = nw__;                                                                                       \
    if (use_ema) (e_) = (e_) - (1.0f - ema_decay) * ((e_) - nw__);                                     \
  } while (0)
  const bool vec = ((s.offset + b0) & 3) == 0 && (s.bf16_offset < 0 || ((s.bf16_offset + b0) & 3) == 0);
  if (vec) {   // 16 bytes per lane and array; scalar tail
    const long long base = s.offset + b0;
    const int n = (int)(b1 - b0), n4 = n >> 2;
    float4* w4 = (float4*)(w + base);
    const float4* g4 = (const float4*)(g + base);
    float4* v4 = (float4*)(v + base);
    float4* e4 = use_ema ? (float4*)(ema + base) : nullptr;
    uint2* h4 = s.bf16_offset >= 0 ? (uint2*)(wbf16 + s.bf16_offset + b0) : nullptr;
    for (int i = threadIdx.x; i < n4; i += OPT_THREADS) {
      float4 wv = w4[i], vv = v4[i], ev = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      const float4 gv = g4[i];
      if (use_ema) ev = e4[i];
      SGD_ELEM(wv.x, gv.x, vv.x, ev.x); SGD_ELEM(wv.y, gv.y, vv.y, ev.y);
      SGD_ELEM(wv.z, gv.z, vv.z, ev.z); SGD_ELEM(wv.w, gv.w, vv.w, ev.w);
      v4[i] = vv;
      w4[i] = wv;
      if (use_ema) e4[i] = ev;
      if (h4) h4[i] = make_uint2((unsigned)rn_f32_to_bf16(wv.x) | ((unsigned)rn_f32_to_bf16(wv.y) << 16),
                                 (unsigned)rn_f32_to_bf16(wv.z) | ((unsigned)rn_f32_to_bf16(wv.w) << 16));
    }
    if ((int)threadIdx.x < (n & 3)) {
      const long long i = b0 + 4 * n4 + threadIdx.x, o = s.offset + i;
      float wv = w[o], vv = v[o], ev = use_ema ? ema[o] : 0.0f;
      SGD_ELEM(wv, g[o], vv, ev);
      v[o] = vv;
      w[o] = wv;
      if (use_ema) ema[o] = ev;
      if (s.bf16_offset >= 0) wbf16[s.bf16_offset + i] = rn_f32_to_bf16(wv);
    }
  } else {
    for (long long i = b0 + threadIdx.x; i < b1; i += OPT_THREADS) {
      const long long o = s.offset + i;
      float wv = w[o], vv = v[o], ev = use_ema ? ema[o] : 0.0f;
      SGD_ELEM(wv, g[o], vv, ev);
      v[o] = vv;
      w[o] = wv;
      if (use_ema) ema[o] = ev;
      if (s.bf16_offset >= 0) wbf16[s.bf16_offset + i] = rn_f32_to_bf16(wv);
    }
  }
#undef SGD_ELEM
}

extern "C" size_t rn_optim_workspace_bytes(int num_blocks, int num_segments) {
  return 2 * rn_align_up((size_t)num_blocks * sizeof(double), 256) + rn_align_up((size_t)num_segments * sizeof(float), 256) + 256;
}
extern "C" int rn_optim_chunk(void) { return OPT_CHUNK; }

struct OptWs { double* partial; double* partial_w; float* factor; };
static OptWs opt_ws(void* workspace, int num_blocks) {
  OptWs w;
  const size_t pb = rn_align_up((size_t)num_blocks * sizeof(double), 256);
  w.partial = (double*)workspace;
  w.partial_w = (double*)((char*)workspace + pb);
  w.factor = (float*)((char*)workspace + 2 * pb);
  return w;
}

// Stage 1 over blocks [block_begin, block_begin + block_count): unscale, weight decay, per-block sums of squares.
// The overlapped all-reduce runs it bucket by bucket as the backward pass completes the buckets.
extern "C" int rn_optim_clip_prepare(float* grads, const float* params, const void* segs_dev,
                                     const int32_t* block_seg_dev, int num_blocks, int block_begin, int block_count,
                                     float wd_coeff, float grad_unscale, float* local_copy, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(grads && params && segs_dev && block_seg_dev && num_blocks > 0 && block_begin >= 0 && block_count >= 0 &&
                   block_begin + block_count <= num_blocks, "rn_optim_clip_prepare: bad argument");
  if (!workspace || workspace_bytes < rn_optim_workspace_bytes(num_blocks, 1)) {
    rn_set_error("rn_optim_clip_prepare: workspace too small");
    return RN_ENOMEM;
  }
  if (block_count == 0) return RN_OK;
  const OptWs w = opt_ws(workspace, num_blocks);
  hipLaunchKernelGGL(optim_sqnorm_kernel, dim3(block_count), dim3(OPT_THREADS), 0, (hipStream_t)stream, grads, params,
                     (const OptSeg*)segs_dev, block_seg_dev, block_begin, wd_coeff, grad_unscale, w.partial, w.partial_w,
                     local_copy);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// Stage 2: per-tensor and global clip factors from the block sums (+ the l2 metric, the "fired" / "not finite" flags)
extern "C" int rn_optim_clip_factors(const void* segs_dev, int num_segments, int num_blocks, float clipnorm,
                                     float wd_alpha, float* metrics, float* flags, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(segs_dev && metrics && num_segments > 0 && num_blocks > 0, "rn_optim_clip_factors: bad argument");
  if (!workspace || workspace_bytes < rn_optim_workspace_bytes(num_blocks, num_segments)) {
    rn_set_error("rn_optim_clip_factors: workspace too small");
    return RN_ENOMEM;
  }
  const OptWs w = opt_ws(workspace, num_blocks);
  hipLaunchKernelGGL(optim_factors_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const OptSeg*)segs_dev,
                     num_segments, w.partial, w.partial_w, clipnorm, wd_alpha, w.factor, metrics, flags);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// Stage 3: grads <- grads * factor in place (correction == NULL), or correction <- (factor - 1) * grads
extern "C" int rn_optim_clip_apply(float* grads, float* correction, const void* segs_dev,
                                   const int32_t* block_seg_dev, int num_blocks, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(grads && segs_dev && block_seg_dev && num_blocks > 0 && workspace, "rn_optim_clip_apply: bad argument");
  const OptWs w = opt_ws(workspace, num_blocks);
  hipLaunchKernelGGL(optim_scale_kernel, dim3(num_blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, grads, correction,
                     (const OptSeg*)segs_dev, block_seg_dev, w.factor);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_optim_clip(float* grads, const float* params, const void* segs_dev, int num_segments,
                             const int32_t* block_seg_dev, int num_blocks, float wd_coeff, float wd_alpha,
                             float grad_unscale, float clipnorm, float* metrics /* dev f32[8] */, void* workspace,
                             size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(grads && params && segs_dev && block_seg_dev && metrics && num_segments > 0 && num_blocks > 0,
               "rn_optim_clip: bad argument");
  int rc = rn_optim_clip_prepare(grads, params, segs_dev, block_seg_dev, num_blocks, 0, num_blocks, wd_coeff,
                                 grad_unscale, nullptr, workspace, workspace_bytes, stream);
  if (rc) return rc;
  rc = rn_optim_clip_factors(segs_dev, num_segments, num_blocks, clipnorm, wd_alpha, metrics, nullptr, workspace,
                             workspace_bytes, stream);
  if (rc) return rc;
  return rn_optim_clip_apply(grads, nullptr, segs_dev, block_seg_dev, num_blocks, workspace, workspace_bytes, stream);
}

extern "C" int rn_optim_sgd_step(float* params, const float* grads, float* momentum_buf, float* ema,
                                 void* params_bf16, const void* segs_dev, const int32_t* block_seg_dev,
                                 int num_blocks, float lr, float momentum, float ema_decay, int nesterov,
                                 const float* skip_flag, void* stream) {
  RN_CHECK_ARG(params && grads && momentum_buf && segs_dev && block_seg_dev && num_blocks > 0,
               "rn_optim_sgd_step: bad argument");
  hipLaunchKernelGGL(optim_sgd_kernel, dim3(num_blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grads,
                     momentum_buf, ema, (uint16_t*)params_bf16, (const OptSeg*)segs_dev, block_seg_dev, lr, momentum,
                     ema_decay, ema ? 1 : 0, nesterov, skip_flag);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- dgrad weight transform: master f32 [Cout][R][S][Cin] -> bf16 [Cin_pad][R][S][Cout], taps
// flipped, so the data gradient is the SAME implicit-GEMM forward kernel run on dy. ---------------
__global__ void __launch_bounds__(256)
pack_dgrad_kernel(const float* __restrict__ w, int R, int S, int Cin, int Cout, int Cin_pad, int Cout_pad,
                  uint16_t* __restrict__ out) {
  const long long total = (long long)Cin_pad * R * S * Cout_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout_pad);
    long long t = i / Cout_pad;
    const int s = (int)(t % S);
    t /= S;
    const int r = (int)(t % R);
    const int ci = (int)(t / R);
    float v = 0.0f;
    if (ci < Cin && co < Cout) v = w[(((long long)co * R + (R - 1 - r)) * S + (S - 1 - s)) * Cin + ci];
    out[i] = rn_f32_to_bf16(v);
  }
}
// every trainable conv's dgrad weights in one launch (one grid row per layer): the training step repacks ~60
// layers after each optimizer step, and 60 launches of a few microseconds each are mostly launch latency
struct DgradPackArgs {
  int n;
  rn_dgrad_pack item[RN_DGRAD_PACK_MAX];
};
// Sub-pixel form of the data gradient of a 3x3 / stride 2 / pad 1 convolution (rn_dgrad_pack.pad_ == 1): dx at
// (2i + a, 2j + b) only sees filter rows r with r = a + 1 - 2(oy - i): phase a = 0 -> r = 1 at dy row i; phase
// a = 1 -> r = 2 at row i and r = 0 at row i + 1 (columns alike).  All four phases share the 2 x 2 window
// {i, i+1} x {j, j+1} of dy, so the whole gradient is ONE stride-1 2x2 convolution of dy with 4 * Cin output channels
// (phase-major) followed by a depth-to-space — 16 tap-channel products per dy pixel instead of the 36 of the
// zero-upsampled form (9 carry data).  Packed bf16 [cout_pad(4*Cin)][2][2][Cout_pad]; row n = (a*2 + b)*Cin + ci.
__device__ __forceinline__ float dgrad_s2_weight(const float* __restrict__ w, int Cin, int Cout, int n, int u, int v, int co) {
  if (n >= 4 * Cin || co >= Cout) return 0.0f;
  const int ph = n / Cin, ci = n - ph * Cin, a = ph >> 1, b = ph & 1;
  const int r = a == 0 ? (u == 0 ? 1 : -1) : (u == 0 ? 2 : 0);
  const int s = b == 0 ? (v == 0 ? 1 : -1) : (v == 0 ? 2 : 0);
  if (r < 0 || s < 0) return 0.0f;
  return w[(((long long)co * 3 + r) * 3 + s) * Cin + ci];
}

__global__ void __launch_bounds__(256) pack_dgrad_batch_kernel(const DgradPackArgs a) {
  const rn_dgrad_pack& it = a.item[blockIdx.y];
  if (it.pad_ == 1) {
    const int Cin = it.Cin, Cout = it.Cout, Cout_pad = it.Cout_pad;
    const int rows = 4 * Cin <= 64 ? 64 : ((4 * Cin + 127) / 128) * 128;
    const long long total = (long long)rows * 4 * Cout_pad;
    uint16_t* __restrict__ out = (uint16_t*)it.w_packed;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
      const int co = (int)(i % Cout_pad);
      const long long t = i / Cout_pad;
      const int v = (int)(t & 1), u = (int)((t >> 1) & 1), n = (int)(t >> 2);
      out[i] = rn_f32_to_bf16(dgrad_s2_weight(it.w_ohwi, Cin, Cout, n, u, v, co));
    }
    return;
  }
  const int R = it.R, S = it.S, Cin = it.Cin, Cout = it.Cout, Cout_pad = it.Cout_pad;
  const int Cin_pad = Cin <= 64 ? 64 : ((Cin + 127) / 128) * 128;   // rn_conv_cout_pad(Cin)
  const long long total = (long long)Cin_pad * R * S * Cout_pad;
  const float* __restrict__ w = it.w_ohwi;
  uint16_t* __restrict__ out = (uint16_t*)it.w_packed;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout_pad);
    long long t = i / Cout_pad;
    const int s = (int)(t % S);
    t /= S;
    const int r = (int)(t % R);
    const int ci = (int)(t / R);
    float v = 0.0f;
    if (ci < Cin && co < Cout) v = w[(((long long)co * R + (R - 1 - r)) * S + (S - 1 - s)) * Cin + ci];
    out[i] = rn_f32_to_bf16(v);
  }
}
extern "C" int rn_pack_conv_weight_dgrad_batch(const rn_dgrad_pack* items, int n, void* stream) {
  RN_CHECK_ARG(items && n >= 0, "rn_pack_conv_weight_dgrad_batch: bad argument");
  for (int base = 0; base < n; base += RN_DGRAD_PACK_MAX) {
    DgradPackArgs a;
    a.n = n - base < RN_DGRAD_PACK_MAX ? n - base : RN_DGRAD_PACK_MAX;
    long long most = 0;
    for (int i = 0; i < a.n; ++i) {
      const rn_dgrad_pack& it = items[base + i];
      RN_CHECK_ARG(it.w_ohwi && it.w_packed && it.R > 0 && it.S > 0 && it.Cin > 0 && it.Cout > 0 &&
                       it.Cout_pad >= it.Cout, "rn_pack_conv_weight_dgrad_batch: bad item %d", base + i);
      RN_CHECK_ARG(it.pad_ == 0 || (it.pad_ == 1 && it.R == 3 && it.S == 3),
                   "rn_pack_conv_weight_dgrad_batch: item %d: the sub-pixel form is for 3x3 kernels", base + i);
      a.item[i] = it;
      const long long total = it.pad_ == 1 ? (long long)rn_conv_cout_pad(4 * it.Cin) * 4 * it.Cout_pad
                                           : (long long)rn_conv_cout_pad(it.Cin) * it.R * it.S * it.Cout_pad;
      if (total > most) most = total;
    }
    const int bx = (int)(rn_cdiv(most, 256) < 256 ? rn_cdiv(most, 256) : 256);
    hipLaunchKernelGGL(pack_dgrad_batch_kernel, dim3(bx, a.n), dim3(256), 0, (hipStream_t)stream, a);
    RN_CHECK_LAUNCH();
  }
  return RN_OK;
}

extern "C" int rn_pack_conv_weight_dgrad(const float* w_ohwi, int R, int S, int Cin, int Cout, int Cout_pad,
                                         void* out, void* stream) {
  RN_CHECK_ARG(w_ohwi && out && R > 0 && S > 0 && Cin > 0 && Cout > 0 && Cout_pad >= Cout,
               "rn_pack_conv_weight_dgrad: bad argument");
  const int Cin_pad = rn_conv_cout_pad(Cin);
  const long long total = (long long)Cin_pad * R * S * Cout_pad;
  int blocks = (int)(rn_cdiv(total, 256) < 4096 ? rn_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(pack_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ohwi, R, S, Cin, Cout,
                     Cin_pad, Cout_pad, (uint16_t*)out);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
