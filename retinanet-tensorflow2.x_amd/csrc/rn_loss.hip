// rn_loss.hip — a9: sigmoid focal loss + Huber box loss, fused forward + backward.
// Reference: retinanet/losses/loss_impl.py:15-28 (FocalLossV1.call), :41-77 (ClassLoss),
//            :88-105 (BoxLoss), retinanet/losses/retinanet_loss.py:37-83 (normaliser, weights).
//
// Per logit x with one-hot label y (targets -1/-2 give an all-zero row, loss_impl.py:52-56):
//   ys = y(1-ls) + 0.5 ls;  ce = max(x,0) - x ys + log1p(exp(-|x|));  p = sigmoid(x)
//   a_t = y==1 ? alpha : 1-alpha;  q = 1 - p_t = y==1 ? 1-p : p;  L = a_t q^gamma ce
//   dL/dx = a_t [ q^gamma (p - ys) + ce * dq^gamma/dx ],
//           dq^gamma/dx = y==1 ? -gamma p q^gamma : gamma (1-p) q^gamma
// masked (L = 0, dL = 0) where the anchor's target is -2 (loss_impl.py:60-70).
// Box: e = pred - target; Huber(delta); weight = (target != 0) per coordinate (:94-101).
// Sums are two-stage and deterministic: per-thread fp32 -> per-block double partial ->
// one finalize block adds the partials in index order.
// Algorithmic bytes 8 B per logit (4 read + 4 grad write) + 4 B/80 target.
#include "rn_common.h"
#include "../../include/rn_math.h"

#define RN_LOSS_THREADS 256
#define RN_LOSS_MAX_LEVELS 8

struct LossLevels {
  int num_levels;
  const float* cls[RN_LOSS_MAX_LEVELS];
  const float* box[RN_LOSS_MAX_LEVELS];
  float* dcls[RN_LOSS_MAX_LEVELS];
  float* dbox[RN_LOSS_MAX_LEVELS];
  // write_grad == 2: gradients as bf16 straight into the padded NHWC dy tensors of the prediction convs'
  // backward pass ([B*H*W][stride] with the level's na*K / na*4 live channels in front)
  uint16_t* dcls16[RN_LOSS_MAX_LEVELS];
  uint16_t* dbox16[RN_LOSS_MAX_LEVELS];
  int cls_stride16, box_stride16, na;
  long long off[RN_LOSS_MAX_LEVELS + 1];   // anchor boundaries
  long long vbeg[RN_LOSS_MAX_LEVELS + 1];  // prefix of B*n_l*(K/V) work items
  int cb[RN_LOSS_MAX_LEVELS + 1];          // focal4_kernel: prefix of the levels' workgroups (a workgroup never crosses a level)
  unsigned chunk;                          // focal4_kernel: work items (4 logits each) per workgroup, a multiple of the block size
};

// The kernel is VALU-bound, not HBM-bound, when exp / log / pow / the divisions are the software routines of
// rn_math.h (~170 instructions per logit: 0.79 ms for 32 x 76725 x 80 logits against ~0.31 ms of HBM time), and the
// loss needs no bit parity with a C restatement — its oracle is float64 numpy with the 1e-5 tolerance of
// north_star.  So: hardware exp2 / log2 / rcp (1 ulp), and log1p(t) for t in (0, 1] as 2 atanh(t / (2 + t)), an odd
// series in s <= 1/3 that is accurate relative to its (possibly tiny) value — a hardware log2(1 + t) is only
// accurate in absolute terms near 1, which is where every confident logit sits.
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float log1p_unit(float t) {   // t in [0, 1]
  const float s = t * hw_rcp(2.0f + t);
  const float z = s * s;
  float p = 1.0f / 15.0f;
  p = fmaf(p, z, 1.0f / 13.0f);
  p = fmaf(p, z, 1.0f / 11.0f);
  p = fmaf(p, z, 1.0f / 9.0f);
  p = fmaf(p, z, 1.0f / 7.0f);
  p = fmaf(p, z, 1.0f / 5.0f);
  p = fmaf(p, z, 1.0f / 3.0f);
  p = fmaf(p, z, 1.0f);
  return 2.0f * s * p;
}

__device__ __forceinline__ void focal_elem(float x, bool pos, float alpha, float gamma, float ls, float& loss,
                                           float& grad) {
  const float y = pos ? 1.0f : 0.0f;
  const float ys = y * (1.0f - ls) + 0.5f * ls;
  const float ax = fabsf(x);
  const float t = hw_exp2(-ax * 1.44269504088896341f);   // e^-|x| in (0,1]
  const float ce = fmaxf(x, 0.0f) - x * ys + log1p_unit(t);
  const float inv = hw_rcp(1.0f + t);
  const float p = x >= 0.0f ? inv : t * inv;        // sigmoid(x), stable both signs
  const float omp = x >= 0.0f ? t * inv : inv;      // 1 - sigmoid(x)
  const float q = pos ? omp : p;
  const float a_t = pos ? alpha : 1.0f - alpha;
  // q^gamma: every shipped config has gamma = 1.5 -> q * sqrt(q), one transcendental instead of log2 + exp2
  const float mod = gamma == 1.5f ? q * __builtin_amdgcn_sqrtf(q) : (q > 0.0f ? hw_exp2(gamma * hw_log2(q)) : 0.0f);
  loss = a_t * mod * ce;
  const float dmod = pos ? -gamma * p * mod : gamma * omp * mod;
  grad = a_t * (mod * (p - ys) + ce * dmod);
}

// ---- K % 4 == 0 (every shipped config: 80 / 90 / 20 classes): four logits per thread and iteration ------------------
// The first version (focal_kernel<4> below, kept for V = 1) found its level with a loop over lv.vbeg and fetched the
// level's pointers / sizes with dynamically indexed loads from the kernel-argument segment IN FRONT of every logit load
// (two dependent memory round trips per iteration), divided three times per iteration, and spent ~62 VALU issue cycles
// per logit: 0.43 ms for the 196 M logits of the bench batch, the sum of its VALU time and its HBM time rather than
// the larger of the two.  Here
//   * a workgroup = (level, chunk of lv.chunk consecutive float4s of that level's [B * n_l][K] logits): everything that
//     depends on the level is uniform and read once; a thread's (image, anchor, class quad, pixel) indices advance
//     incrementally by the block stride (no division inside the loop);
//   * the logits / targets of iteration i + 1 are loaded before iteration i is computed;
//   * two logits at a time through the packed fp32 pipe (v_pk_mul / add / fma_f32: the log1p series, the products of the
//     loss and its gradient — 2 results per issue cycle), ONE reciprocal r = 1 / ((1 + t)(2 + t)) for both
//     1 / (1 + t) = r (2 + t) (the sigmoid) and 1 / (2 + t) = r (1 + t) (the log1p argument).
typedef float loss_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ loss_f2 f2_fma(loss_f2 a, loss_f2 b, loss_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ loss_f2 f2_splat(float v) { return loss_f2{v, v}; }

template <bool G15>
__device__ __forceinline__ void focal_pair(loss_f2 x, bool pos0, bool pos1, float alpha, float gamma, float ls,
                                           loss_f2& loss, loss_f2& grad) {
  const loss_f2 y = {pos0 ? 1.0f : 0.0f, pos1 ? 1.0f : 0.0f};
  const loss_f2 ys = y * f2_splat(1.0f - ls) + f2_splat(0.5f * ls);
  const loss_f2 t = {hw_exp2(-fabsf(x.x) * 1.44269504088896341f), hw_exp2(-fabsf(x.y) * 1.44269504088896341f)};
  const loss_f2 xp = {fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f)};
  const loss_f2 d1 = f2_splat(1.0f) + t, d2 = f2_splat(2.0f) + t;
  const loss_f2 dd = d1 * d2;                       // in (2, 6]
  const loss_f2 r = {hw_rcp(dd.x), hw_rcp(dd.y)};
  const loss_f2 inv = r * d2;                       // 1 / (1 + t)
  const loss_f2 s = t * (r * d1);                   // t / (2 + t) <= 1/3
  const loss_f2 z = s * s;
  loss_f2 pl = f2_splat(1.0f / 15.0f);              // log1p(t) = 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ...)
  pl = f2_fma(pl, z, f2_splat(1.0f / 13.0f));
  pl = f2_fma(pl, z, f2_splat(1.0f / 11.0f));
  pl = f2_fma(pl, z, f2_splat(1.0f / 9.0f));
  pl = f2_fma(pl, z, f2_splat(1.0f / 7.0f));
  pl = f2_fma(pl, z, f2_splat(1.0f / 5.0f));
  pl = f2_fma(pl, z, f2_splat(1.0f / 3.0f));
  pl = f2_fma(pl, z, f2_splat(1.0f));
  const loss_f2 l1p = f2_splat(2.0f) * s * pl;
  const loss_f2 ce = xp - x * ys + l1p;
  const loss_f2 ti = t * inv;
  const loss_f2 p = {x.x >= 0.0f ? inv.x : ti.x, x.y >= 0.0f ? inv.y : ti.y};       // sigmoid(x), stable both signs
  const loss_f2 omp = {x.x >= 0.0f ? ti.x : inv.x, x.y >= 0.0f ? ti.y : inv.y};     // 1 - sigmoid(x)
  const loss_f2 q = {pos0 ? omp.x : p.x, pos1 ? omp.y : p.y};
  const loss_f2 a_t = {pos0 ? alpha : 1.0f - alpha, pos1 ? alpha : 1.0f - alpha};
  loss_f2 mod;
  if (G15) {   // q^1.5 = q sqrt(q): one transcendental instead of log2 + exp2
    mod = q * loss_f2{__builtin_amdgcn_sqrtf(q.x), __builtin_amdgcn_sqrtf(q.y)};
  } else {
    mod.x = q.x > 0.0f ? hw_exp2(gamma * hw_log2(q.x)) : 0.0f;
    mod.y = q.y > 0.0f ? hw_exp2(gamma * hw_log2(q.y)) : 0.0f;
  }
  loss = a_t * mod * ce;
  const loss_f2 dm = {pos0 ? -p.x : omp.x, pos1 ? -p.y : omp.y};
  const loss_f2 dmod = f2_splat(gamma) * dm * mod;
  grad = a_t * (mod * (p - ys) + ce * dmod);
}

template <bool G15>
__global__ void __launch_bounds__(RN_LOSS_THREADS)
focal4_kernel(LossLevels lv, int K, long long A, const float* __restrict__ class_targets,
              const float* __restrict__ normalizer, float alpha, float gamma, float ls, float gscale,
              int write_grad, double* __restrict__ partials) {
  int l = 0;
  while (l + 1 < lv.num_levels && (int)blockIdx.x >= lv.cb[l + 1]) ++l;
  const unsigned KV = (unsigned)K >> 2;
  const unsigned n_l = (unsigned)(lv.off[l + 1] - lv.off[l]);
  const unsigned items = (unsigned)(lv.vbeg[l + 1] - lv.vbeg[l]);   // B * n_l * KV < 2^32 (the launcher checks)
  const unsigned start = ((unsigned)blockIdx.x - (unsigned)lv.cb[l]) * lv.chunk;
  const unsigned end = items - start > lv.chunk ? start + lv.chunk : items;
  const float4* __restrict__ src = (const float4*)lv.cls[l];
  const float* __restrict__ tgt = class_targets + lv.off[l];
  const unsigned PK = (unsigned)lv.na * KV;                           // float4s per pixel (write_grad == 2)
  // block stride in (row, class quad) and (pixel, quad inside the pixel) steps
  const unsigned dq = RN_LOSS_THREADS / KV, dr = RN_LOSS_THREADS - dq * KV;
  const unsigned pq = write_grad == 2 ? RN_LOSS_THREADS / PK : 0, pr = write_grad == 2 ? RN_LOSS_THREADS - pq * PK : 0;
  const float gs = gscale / normalizer[0];
  unsigned local = start + threadIdx.x;
  unsigned row = local / KV, kv = local - row * KV;
  unsigned b = row / n_l, j = row - b * n_l;
  unsigned pix = 0, prem = 0;
  if (write_grad == 2) { pix = local / PK; prem = local - pix * PK; }
  float acc = 0.0f;
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float ct = -1.0f;
  if (local < end) { v = src[local]; ct = tgt[(long long)b * A + j]; }
  while (local < end) {
    // the next iteration's indices and loads first
    const unsigned nlocal = local + RN_LOSS_THREADS;
    unsigned nkv = kv + dr, nj = j + dq, nb = b;
    if (nkv >= KV) { nkv -= KV; ++nj; }
    while (nj >= n_l) { nj -= n_l; ++nb; }
    float4 vn = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float ctn = -1.0f;
    if (nlocal < end) { vn = src[nlocal]; ctn = tgt[(long long)nb * A + nj]; }
    const bool ignore = (ct == -2.0f);
    const int c0 = (int)ct - (int)(kv * 4);     // the positive class relative to this quad
    loss_f2 lo0, gr0, lo1, gr1;
    focal_pair<G15>(loss_f2{v.x, v.y}, c0 == 0, c0 == 1, alpha, gamma, ls, lo0, gr0);
    focal_pair<G15>(loss_f2{v.z, v.w}, c0 == 2, c0 == 3, alpha, gamma, ls, lo1, gr1);
    if (ignore) { lo0 = f2_splat(0.0f); lo1 = lo0; gr0 = lo0; gr1 = lo0; }
    acc += lo0.x; acc += lo0.y; acc += lo1.x; acc += lo1.y;
    gr0 = gr0 * f2_splat(gs);
    gr1 = gr1 * f2_splat(gs);
    if (write_grad == 1) {
      ((float4*)lv.dcls[l])[local] = make_float4(gr0.x, gr0.y, gr1.x, gr1.y);
    } else if (write_grad == 2) {
      uint16_t* dst = lv.dcls16[l] + (long long)pix * lv.cls_stride16 + prem * 4;
      *(uint2*)dst = make_uint2(rn_pack_bf16x2(gr0.x, gr0.y), rn_pack_bf16x2(gr1.x, gr1.y));
      prem += pr; pix += pq;
      if (prem >= PK) { prem -= PK; ++pix; }
    }
    local = nlocal; kv = nkv; j = nj; b = nb; v = vn; ct = ctn;
  }
  __shared__ double sred[RN_LOSS_THREADS / 64];
  double d = rn_wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < RN_LOSS_THREADS / 64; ++w) s += sred[w];
    partials[blockIdx.x] = s;
  }
}


template <int V>
__global__ void __launch_bounds__(RN_LOSS_THREADS)
focal_kernel(LossLevels lv, int B, int K, long long A, const float* __restrict__ class_targets,
             const float* __restrict__ normalizer, float alpha, float gamma, float ls, float gscale,
             int write_grad, double* __restrict__ partials) {
  const int KV = K / V;
  const long long total = lv.vbeg[lv.num_levels];
  float acc = 0.0f;
  const float gs = gscale / normalizer[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < lv.num_levels && i >= lv.vbeg[l + 1]) ++l;
    // 32-bit index arithmetic inside a level (the launcher checks B * n_l * K / V < 2^32): the 64-bit divisions
    // this replaced cost ~120 VALU instructions each, per 4 logits — the kernel was division-bound, not HBM-bound
    const unsigned local = (unsigned)(i - lv.vbeg[l]);
    const unsigned n_l = (unsigned)(lv.off[l + 1] - lv.off[l]);
    const unsigned row = local / (unsigned)KV;  // b * n_l + j
    const int kv = (int)(local - row * (unsigned)KV);
    const unsigned b = row / n_l;
    const unsigned j = row - b * n_l;
    const float ct = class_targets[(long long)b * A + lv.off[l] + j];
    const bool ignore = (ct == -2.0f);
    const int cls = (int)ct;
    const float* src = lv.cls[l] + (long long)row * K + (long long)kv * V;
    float x[V], g[V];
    if (V == 4) {
      const float4 v = *(const float4*)src;
      x[0] = v.x; x[1 % V] = v.y; x[2 % V] = v.z; x[3 % V] = v.w;
    } else {
      x[0] = src[0];
    }
#pragma unroll
    for (int u = 0; u < V; ++u) {
      float lo, gr;
      focal_elem(x[u], (kv * V + u) == cls, alpha, gamma, ls, lo, gr);
      if (ignore) { lo = 0.0f; gr = 0.0f; }
      acc += lo;
      g[u] = gr * gs;
    }
    if (write_grad == 1) {
      float* dst = lv.dcls[l] + (long long)row * K + (long long)kv * V;
      if (V == 4) *(float4*)dst = make_float4(g[0], g[1 % V], g[2 % V], g[3 % V]);
      else dst[0] = g[0];
    } else if (write_grad == 2) {
      const unsigned pix = row / (unsigned)lv.na;                // rows are (image, anchor): na anchors per pixel
      const unsigned ch = (row - pix * (unsigned)lv.na) * (unsigned)K + (unsigned)(kv * V);
      uint16_t* dst = lv.dcls16[l] + (long long)pix * lv.cls_stride16 + ch;
      if (V == 4) *(uint2*)dst = make_uint2(rn_pack_bf16x2(g[0], g[1 % V]), rn_pack_bf16x2(g[2 % V], g[3 % V]));
      else dst[0] = rn_f32_to_bf16(g[0]);
    }
  }
  __shared__ double sred[RN_LOSS_THREADS / 64];
  double d = rn_wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < RN_LOSS_THREADS / 64; ++w) s += sred[w];
    partials[blockIdx.x] = s;
  }
}

__global__ void __launch_bounds__(RN_LOSS_THREADS)
huber_kernel(LossLevels lv, int B, long long A, const float4* __restrict__ box_targets,
             const float* __restrict__ normalizer, float delta, float gscale, int write_grad,
             double* __restrict__ partials) {
  const long long total = (long long)B * A;
  float acc = 0.0f;
  const float gs = gscale / normalizer[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / A;
    const long long a = i - b * A;
    int l = 0;
    while (l + 1 < lv.num_levels && a >= lv.off[l + 1]) ++l;
    const long long n_l = lv.off[l + 1] - lv.off[l];
    const long long row = b * n_l + (a - lv.off[l]);
    const float4 p = ((const float4*)lv.box[l])[row];
    const float4 t = box_targets[i];
    const float pv[4] = {p.x, p.y, p.z, p.w};
    const float tv[4] = {t.x, t.y, t.z, t.w};
    float gv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float e = pv[c] - tv[c];
      const float ae = fabsf(e);
      const bool quad = ae <= delta;
      float lo = quad ? 0.5f * (e * e) : delta * ae - 0.5f * (delta * delta);
      float gr = quad ? e : (e > 0.0f ? delta : -delta);
      if (tv[c] == 0.0f) { lo = 0.0f; gr = 0.0f; }
      acc += lo;
      gv[c] = gr * gs;
    }
    if (write_grad == 1) {
      ((float4*)lv.dbox[l])[row] = make_float4(gv[0], gv[1], gv[2], gv[3]);
    } else if (write_grad == 2) {
      const long long pix = row / lv.na;
      const int ch = (int)(row - pix * lv.na) * 4;
      *(uint2*)(lv.dbox16[l] + pix * lv.box_stride16 + ch) =
          make_uint2(rn_pack_bf16x2(gv[0], gv[1]), rn_pack_bf16x2(gv[2], gv[3]));
    }
  }
  __shared__ double sred[RN_LOSS_THREADS / 64];
  double d = rn_wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < RN_LOSS_THREADS / 64; ++w) s += sred[w];
    partials[blockIdx.x] = s;
  }
}

// one wavefront: lane j adds partials j, j+64, ... in order, then a fixed-shape butterfly adds the
// 64 lane sums — deterministic, and ~40x shorter than one thread walking 4096 doubles
__global__ void loss_finalize(const double* __restrict__ cls_part, int ncls, const double* __restrict__ box_part,
                              int nbox, const float* __restrict__ normalizer, float box_w, float cls_w,
                              float* __restrict__ losses) {
  double cs = 0.0, bs = 0.0;
  for (int i = threadIdx.x; i < ncls; i += 64) cs += cls_part[i];
  for (int i = threadIdx.x; i < nbox; i += 64) bs += box_part[i];
  cs = rn_wave_sum_d(cs);
  bs = rn_wave_sum_d(bs);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float norm = normalizer[0];
    const float class_loss = (float)cs / norm;
    const float box_loss = ((float)bs / 4.0f) / norm;  // loss_impl.py:105 then retinanet_loss.py:59-60
    losses[0] = box_loss;
    losses[1] = class_loss;
    losses[2] = box_w * box_loss + cls_w * class_loss;
    losses[3] = norm;
  }
}

static int loss_blocks(long long items) {
  long long b = rn_cdiv(items, RN_LOSS_THREADS);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" size_t rn_loss_workspace_bytes(int B, int64_t A, int K) {
  (void)B; (void)A; (void)K;
  return 2 * 2048 * sizeof(double);
}

static int loss_launch(const float* const* class_logits, const float* const* box_preds,
                       float* const* d_class_logits, float* const* d_box_preds, void* const* d_class_bf16,
                       void* const* d_box_bf16, int class_pix_stride, int box_pix_stride, int anchors_per_location,
                       const int64_t* level_offsets, int num_levels, int B, int K,
                       const float* class_targets, const float* box_targets,
                       const float* normalizer, float alpha, float gamma,
                       float label_smoothing, float delta, float box_loss_weight,
                       float class_loss_weight, float grad_scale, float* losses,
                       void* workspace, size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(class_logits && box_preds && level_offsets && class_targets && box_targets && normalizer && losses,
               "rn_retinanet_loss_fwd_bwd: null argument");
  RN_CHECK_ARG(num_levels >= 1 && num_levels <= RN_LOSS_MAX_LEVELS && B > 0 && K > 0,
               "rn_retinanet_loss_fwd_bwd: bad num_levels=%d B=%d K=%d", num_levels, B, K);
  if (!workspace || workspace_bytes < rn_loss_workspace_bytes(B, 0, K)) {
    rn_set_error("rn_retinanet_loss_fwd_bwd: workspace too small");
    return RN_ENOMEM;
  }
  const int V = (K % 4 == 0) ? 4 : 1;
  const int write_grad = (d_class_bf16 && d_box_bf16) ? 2 : (d_class_logits && d_box_preds) ? 1 : 0;
  LossLevels lv = {};
  lv.cls_stride16 = class_pix_stride; lv.box_stride16 = box_pix_stride; lv.na = anchors_per_location;
  if (write_grad == 2) {
    RN_CHECK_ARG(anchors_per_location > 0 && class_pix_stride >= anchors_per_location * K &&
                 box_pix_stride >= anchors_per_location * 4 && class_pix_stride % 4 == 0 && box_pix_stride % 4 == 0,
                 "rn_retinanet_loss_fwd_bwd_bf16: bad strides %d / %d for %d anchors per location", class_pix_stride,
                 box_pix_stride, anchors_per_location);
  }
  lv.num_levels = num_levels;
  lv.off[0] = level_offsets[0];
  lv.vbeg[0] = 0;
  for (int l = 0; l < num_levels; ++l) {
    lv.cls[l] = class_logits[l];
    lv.box[l] = box_preds[l];
    lv.dcls[l] = write_grad == 1 ? d_class_logits[l] : nullptr;
    lv.dbox[l] = write_grad == 1 ? d_box_preds[l] : nullptr;
    lv.dcls16[l] = write_grad == 2 ? (uint16_t*)d_class_bf16[l] : nullptr;
    lv.dbox16[l] = write_grad == 2 ? (uint16_t*)d_box_bf16[l] : nullptr;
    if (write_grad == 2)
      RN_CHECK_ARG(lv.dcls16[l] && lv.dbox16[l] && (level_offsets[l + 1] - level_offsets[l]) % anchors_per_location == 0,
                   "rn_retinanet_loss_fwd_bwd_bf16: level %d: null output or anchors not a multiple of %d", l,
                   anchors_per_location);
    lv.off[l + 1] = level_offsets[l + 1];
    const long long n_l = level_offsets[l + 1] - level_offsets[l];
    RN_CHECK_ARG(n_l > 0, "rn_retinanet_loss_fwd_bwd: empty level %d", l);
    lv.vbeg[l + 1] = lv.vbeg[l] + (long long)B * n_l * (K / V);
  }
  RN_CHECK_ARG(lv.off[0] == 0, "rn_retinanet_loss_fwd_bwd: level_offsets[0] must be 0");
  const long long A = lv.off[num_levels];
  hipStream_t st = (hipStream_t)stream;
  double* cls_part = (double*)workspace;
  double* box_part = cls_part + 2048;
  for (int l = 0; l < num_levels; ++l)
    RN_CHECK_ARG(lv.vbeg[l + 1] - lv.vbeg[l] < (1ll << 32), "rn_retinanet_loss: level %d has >= 2^32 logit vectors", l);
  int nb_cls = loss_blocks(lv.vbeg[num_levels]);
  const int nb_box = loss_blocks((long long)B * A);
  if (V == 4) {
    // workgroup = (level, chunk): the chunk is the smallest multiple of the block size that keeps the launch inside the
    // 2048 partial sums of the workspace
    long long iters = rn_cdiv(lv.vbeg[num_levels], 2000ll * RN_LOSS_THREADS);
    if (iters < 1) iters = 1;
    for (;; ++iters) {
      long long nb = 0;
      lv.cb[0] = 0;
      for (int l = 0; l < num_levels; ++l) {
        nb += rn_cdiv(lv.vbeg[l + 1] - lv.vbeg[l], iters * RN_LOSS_THREADS);
        lv.cb[l + 1] = (int)nb;
      }
      if (nb <= 2048) break;
    }
    RN_CHECK_ARG(iters * RN_LOSS_THREADS < (1ll << 31), "rn_retinanet_loss: too many logits");
    for (int l = 0; l < num_levels; ++l)   // a thread's index may run one block stride past the level's end
      RN_CHECK_ARG(lv.vbeg[l + 1] - lv.vbeg[l] < (1ll << 32) - (1ll << 16), "rn_retinanet_loss: level %d has too many logits", l);
    lv.chunk = (unsigned)(iters * RN_LOSS_THREADS);
    nb_cls = lv.cb[num_levels];
    if (gamma == 1.5f)
      hipLaunchKernelGGL(focal4_kernel<true>, dim3(nb_cls), dim3(RN_LOSS_THREADS), 0, st, lv, K, A, class_targets,
                         normalizer, alpha, gamma, label_smoothing, class_loss_weight * grad_scale, write_grad, cls_part);
    else
      hipLaunchKernelGGL(focal4_kernel<false>, dim3(nb_cls), dim3(RN_LOSS_THREADS), 0, st, lv, K, A, class_targets,
                         normalizer, alpha, gamma, label_smoothing, class_loss_weight * grad_scale, write_grad, cls_part);
  } else {
    hipLaunchKernelGGL(focal_kernel<1>, dim3(nb_cls), dim3(RN_LOSS_THREADS), 0, st, lv, B, K, A, class_targets,
                       normalizer, alpha, gamma, label_smoothing, class_loss_weight * grad_scale, write_grad,
                       cls_part);
  }
  RN_CHECK_LAUNCH();
  hipLaunchKernelGGL(huber_kernel, dim3(nb_box), dim3(RN_LOSS_THREADS), 0, st, lv, B, A,
                     (const float4*)box_targets, normalizer, delta, box_loss_weight * grad_scale / 4.0f,
                     write_grad, box_part);
  RN_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(64), 0, st, cls_part, nb_cls, box_part, nb_box, normalizer,
                     box_loss_weight, class_loss_weight, losses);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_retinanet_loss_fwd_bwd(const float* const* class_logits, const float* const* box_preds,
                                         float* const* d_class_logits, float* const* d_box_preds,
                                         const int64_t* level_offsets, int num_levels, int B, int K,
                                         const float* class_targets, const float* box_targets,
                                         const float* normalizer, float alpha, float gamma,
                                         float label_smoothing, float delta, float box_loss_weight,
                                         float class_loss_weight, float grad_scale, float* losses,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return loss_launch(class_logits, box_preds, d_class_logits, d_box_preds, nullptr, nullptr, 0, 0, 0, level_offsets,
                     num_levels, B, K, class_targets, box_targets, normalizer, alpha, gamma, label_smoothing, delta,
                     box_loss_weight, class_loss_weight, grad_scale, losses, workspace, workspace_bytes, stream);
}

extern "C" int rn_retinanet_loss_fwd_bwd_bf16(const float* const* class_logits, const float* const* box_preds,
                                              void* const* d_class_bf16, void* const* d_box_bf16,
                                              int class_pix_stride, int box_pix_stride, int anchors_per_location,
                                              const int64_t* level_offsets, int num_levels, int B, int K,
                                              const float* class_targets, const float* box_targets,
                                              const float* normalizer, float alpha, float gamma,
                                              float label_smoothing, float delta, float box_loss_weight,
                                              float class_loss_weight, float grad_scale, float* losses,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(d_class_bf16 && d_box_bf16, "rn_retinanet_loss_fwd_bwd_bf16: null gradient outputs");
  return loss_launch(class_logits, box_preds, nullptr, nullptr, d_class_bf16, d_box_bf16, class_pix_stride,
                     box_pix_stride, anchors_per_location, level_offsets, num_levels, B, K, class_targets, box_targets,
                     normalizer, alpha, gamma, label_smoothing, delta, box_loss_weight, class_loss_weight, grad_scale,
                     losses, workspace, workspace_bytes, stream);
}
