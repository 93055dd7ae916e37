// rn_train.hip — training-mode BatchNorm (forward + backward), bias gradients and the
// backward of the pooling / FPN-fusion / BalanceFeatures layers.  All activations and
// activation gradients are bf16 [P][C] (P = N*H*W pixels, NHWC), math is fp32, 8 channels
// (16 bytes) per thread.  HBM-bound.
//
// Reference: tf.keras BatchNormalization / SyncBatchNormalization selected by
// retinanet/model/utils.py:7-22 (momentum/epsilon from the config), used at
// resnet.py:59-79, fpn_base.py:51-52, fpn.py:54-69, detection_head.py:68-74,99; the
// activation (model/utils.py:45-70) and the residual add (resnet.py:248) are fused in.
// Semantics restated from TF 2.8: normalise with the biased batch
// variance; moving stats <- m*mom + batch*(1-mom), with the Bessel-corrected variance for the
// fused single-replica layer and the biased one for SyncBatchNormalization.
//
// Two-stage deterministic reductions: stage 1 writes per-(row chunk) partial sums, stage 2
// (the finalize kernels) adds them in index order in double.  SyncBN inserts an all-reduce of
// the [2][C] sums between the stages (host side, RCCL).
#include "rn_common.h"

#define TR_THREADS 256

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
__device__ __forceinline__ float act_mask(float z, int act) {
  if (act == RN_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
  if (act == RN_ACT_RELU6) return (z > 0.0f && z < 6.0f) ? 1.0f : 0.0f;
  return 1.0f;
}
// relu / relu6 pass-through mask from the pre-activation u = y*scale + shift (layers without a residual
// input): tf.nn.relu / relu6 gradients gate on the op's bf16 INPUT, i.e. on round_bf16(u) — so the stored
// output z does not have to be read back (saves one tensor read in each of the two backward passes).
__device__ __forceinline__ float act_mask_u(float u, int act) {
  if (act == RN_ACT_RELU) return u > 0.0f ? 1.0f : 0.0f;      // rounding to bf16 never changes the sign
  if (act == RN_ACT_RELU6) {
    const float ub = rn_rb(u);                                 // v_cvt_pk_bf16_f32 (RNE)
    return (ub > 0.0f && ub < 6.0f) ? 1.0f : 0.0f;
  }
  return 1.0f;
}
// d act(u) / du: relu / relu6 from the stored output z, swish from the recomputed pre-activation u
// (swish'(u) = s + u*s*(1-s), s = sigmoid(u); tf.nn.swish's registered gradient)
__device__ __forceinline__ float act_deriv(float z, float u, int act) {
  if (act == RN_ACT_SWISH) {
    const float sg = 1.0f / (1.0f + __expf(-u));
    return sg + u * sg * (1.0f - sg);
  }
  return act_mask(z, act);
}

// How the BatchNorm backward kernels turn dz into g = dz * act'(.): decided on the host once per launch and
// compiled in (a runtime `act` / `residual` test inside the unrolled 8-channel loops becomes a chain of scalar
// compares and branches PER ELEMENT — the passes are then issue-bound, not HBM-bound).
// G_MASK: relu / relu6 behind a residual add with the gate stored by the forward pass as ONE BIT per element
// (rn_bn_segment.act_mask): the backward passes read P*C/8 bytes instead of the 2*P*C of z.
enum { G_GENERIC = 0, G_NONE = 1, G_U_RELU = 2, G_U_RELU6 = 3, G_Z_RELU = 4, G_Z_RELU6 = 5, G_SWISH = 6, G_MASK = 7 };
template <int G>
__device__ __forceinline__ float grad_gate(float dz, float z, float u, int act, bool from_u) {
  if (G == G_NONE || G == G_MASK) return dz;                     // G_MASK: the caller applies the stored bit
  if (G == G_U_RELU) return u > 0.0f ? dz : 0.0f;               // rounding to bf16 never changes the sign
  if (G == G_U_RELU6) {
    const float ub = rn_rb(u);                                   // the op's bf16 input (see act_mask_u)
    return (ub > 0.0f && ub < 6.0f) ? dz : 0.0f;
  }
  if (G == G_Z_RELU) return z > 0.0f ? dz : 0.0f;
  if (G == G_Z_RELU6) return (z > 0.0f && z < 6.0f) ? dz : 0.0f;
  if (G == G_SWISH) {
    // v_rcp_f32 (1 ulp) instead of the correctly rounded division the build flags give `/` (~10 instructions): the
    // swish passes of EfficientNet are VALU-bound, and the gate is a factor of a gradient that is rounded to 16 bits
    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    return dz * (sg + u * sg * (1.0f - sg));
  }
  return act == RN_ACT_NONE ? dz : dz * (from_u ? act_mask_u(u, act) : act_deriv(z, u, act));
}
// the gate mode of a problem: one mode when every segment agrees on "has a residual input", else generic
static int bn_gate_mode(const rn_bn_problem* p) {
  if (p->act == RN_ACT_NONE) return G_NONE;
  if (p->act == RN_ACT_SWISH) return G_SWISH;
  int with_res = 0;
  for (int i = 0; i < p->num_segments; ++i) with_res += p->seg[i].residual ? 1 : 0;
  if (with_res != 0 && with_res != p->num_segments) return G_GENERIC;
  if (with_res == 0) return p->act == RN_ACT_RELU ? G_U_RELU : (p->act == RN_ACT_RELU6 ? G_U_RELU6 : G_GENERIC);
  int with_mask = 0;
  for (int i = 0; i < p->num_segments; ++i) with_mask += p->seg[i].act_mask ? 1 : 0;
  if (with_mask == p->num_segments && (p->act == RN_ACT_RELU || p->act == RN_ACT_RELU6)) return G_MASK;
  return p->act == RN_ACT_RELU ? G_Z_RELU : (p->act == RN_ACT_RELU6 ? G_Z_RELU6 : G_GENERIC);
}
#define BN_DISPATCH_GATE(mode_, CALL_)                  \
  switch (mode_) {                                      \
    case G_NONE: CALL_(G_NONE); break;                  \
    case G_U_RELU: CALL_(G_U_RELU); break;              \
    case G_U_RELU6: CALL_(G_U_RELU6); break;            \
    case G_Z_RELU: CALL_(G_Z_RELU); break;              \
    case G_Z_RELU6: CALL_(G_Z_RELU6); break;            \
    case G_SWISH: CALL_(G_SWISH); break;                \
    case G_MASK: CALL_(G_MASK); break;                  \
    default: CALL_(G_GENERIC); break;                   \
  }

static int tr_blocks(long long items, int cap = 8192) {
  long long b = rn_cdiv(items, TR_THREADS);
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

struct BnSegDev {
  const uint4* y; uint4* z; const uint4* residual; const uint4* dz; uint4* dy; uint4* dres;
  float* sums; float* fwd; float* bsums;
  const float* gamma; const float* beta; float* moving_mean; float* moving_var; float* dgamma; float* dbeta;
  long long P;
  int C, dres_accumulate, chunks, rows_per_chunk;
  int slab_groups, nslab;   // bn_colreduce_kernel: 8-channel groups per workgroup slab, slabs per row (bn_slab_plan)
  const float* sample_scale; long long rows_per_sample;
  unsigned char* mask;   // one bit per element: act'(z) != 0 (written by bn_apply, read by the G_MASK backward passes)
  float* colsum;         // rn_bn_segment.dy_colsum_partial: per-chunk column sums of the stored dy (bn_bwd_apply), or null
};
struct BnArgs {
  int nseg, act, bessel, mode, fuse_finalize;
  int chunk_u;          // elementwise passes: 0 = grid-stride lanes; U > 0 = one workgroup per 256 * U contiguous 16-byte units
  float eps, momentum, count_scale;
  float* ws;           // partials: [seg][chunk][2][C] laid out with ws_off
  long long ws_off[RN_CONV_MAX_SEGMENTS];
  // the split form of the stage-2 reduction (bn_colreduce_final_split_kernel): parts per segment (1 = not split), byte
  // offsets of the segment's ticket counters / part slots in the workspace tail, total workspace bytes of this mode
  int ks[RN_CONV_MAX_SEGMENTS];
  long long cnt_off[RN_CONV_MAX_SEGMENTS], slot_off[RN_CONV_MAX_SEGMENTS], total_bytes;
  BnSegDev seg[RN_CONV_MAX_SEGMENTS];
};
#define BN_KS_MAX 32        // parts of a split stage-2 reduction
#define BN_KS_CHUNKS 512    // segments with more partial rows than this are split, ~256 rows per part

// mode 0: (sum y, sum y^2); mode 1: (sum g, sum g*xhat), g = dz*mask(z), xhat = (y-mean)*invstd
// A workgroup = G channel groups (8 channels, 16 bytes each) x RL = 256 / G row lanes of one slab of the rows of its chunk.
// G is chosen per segment (bn_slab_plan): until round 6 it was 8 — 64-channel slabs — and the channel counts of
// EfficientNet (144, 288, 816, 1 392; 24 ... 232 behind the projection convs) left the last slab of every row with 25 - 75 %
// of its lanes dead: the swish form of this kernel is VALU-bound (8 v_exp + 8 v_rcp + ~130 other instructions per 16 bytes
// of y and dz), so dead lanes were time.
// SGT = 8: every segment of the launch has 64-channel slabs (ResNet: the index arithmetic folds to shifts, as before
// round 6); SGT = 0: per segment at run time.
template <int G, int SGT = 0>   // G < 0: mode 0 (forward statistics); else the gradient gate of mode 1
__global__ void __launch_bounds__(TR_THREADS) bn_colreduce_kernel(const BnArgs a) {
  const BnSegDev& s = a.seg[blockIdx.z];
  const int chunk = blockIdx.x;
  const int slab = blockIdx.y;
  if (chunk >= s.chunks || slab >= s.nslab) return;
  const int C8 = s.C >> 3;
  const int SG = SGT ? SGT : s.slab_groups, RL = TR_THREADS / SG;
  const int rl = threadIdx.x / SG, cg = threadIdx.x - rl * SG;
  const int c8 = slab * SG + cg;
  const bool live = c8 < C8 && rl < RL;
  const long long r0 = (long long)chunk * s.rows_per_chunk;
  long long r1 = r0 + s.rows_per_chunk;
  if (r1 > s.P) r1 = s.P;
  float s0[8], s1[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.0f;
  float mean[8], istd[8], scq[8], shq[8];
  if (G >= 0 && live) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      mean[q] = s.fwd[0 * s.C + c8 * 8 + q];
      istd[q] = s.fwd[1 * s.C + c8 * 8 + q];
      scq[q] = s.fwd[2 * s.C + c8 * 8 + q];
      shq[q] = s.fwd[3 * s.C + c8 * 8 + q];
    }
  }
  if (live) {
    for (long long r = r0 + rl; r < r1; r += RL) {
      const long long o = r * C8 + c8;
      const bf8 y = unpack8(s.y[o]);
      if (G < 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          s0[q] += y.v[q];
          s1[q] += y.v[q] * y.v[q];
        }
      } else {
        const bf8 dz = unpack8(s.dz[o]);
        const bool from_u = !s.residual && (a.act == RN_ACT_RELU || a.act == RN_ACT_RELU6);
        constexpr bool need_z = G == G_Z_RELU || G == G_Z_RELU6 || G == G_GENERIC;
        bf8 z;
        if (need_z && (G != G_GENERIC || (a.act != RN_ACT_NONE && !from_u))) z = unpack8(s.z[o]);
        unsigned bits = 0xffu;
        if (G == G_MASK) bits = s.mask[o];
        float m = 1.0f;
        if (s.sample_scale) m = s.sample_scale[(int)r / (int)s.rows_per_sample];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float u = y.v[q] * scq[q] + shq[q];
          float g = grad_gate<G < 0 ? 0 : G>(dz.v[q], z.v[q], u, a.act, from_u) * m;
          if (G == G_MASK) g = ((bits >> q) & 1u) ? g : 0.0f;
          s0[q] += g;
          s1[q] += g * ((y.v[q] - mean[q]) * istd[q]);
        }
      }
    }
  }
  // [which][row lane][SG * 8 + 1]: RL * (SG * 8 + 1) <= 256 / SG * (SG * 8 + 1) <= 2 048 + 256 floats per plane
  __shared__ float red[SGT == 8 ? 2 * 32 * 65 : 2 * (TR_THREADS * 8 + TR_THREADS)];
  const int W = SG * 8 + 1;
  if (rl < RL) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      red[rl * W + cg * 8 + q] = s0[q];
      red[RL * W + rl * W + cg * 8 + q] = s1[q];
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 2 * SG * 8; t += TR_THREADS) {
    const int which = t >= SG * 8 ? 1 : 0, c = t - which * SG * 8;
    float v = 0.0f;
    for (int r = 0; r < RL; ++r) v += red[which * RL * W + r * W + c];
    const int ch = slab * SG * 8 + c;
    if (ch < s.C) a.ws[a.ws_off[blockIdx.z] + ((long long)chunk * 2 + which) * s.C + ch] = v;
  }
}

// groups per slab / slabs per row of a segment with C8 = C / 8 channel groups and `chunks` row chunks.  The 64-channel
// slabs (8 groups x 32 row lanes) stay wherever they keep >= 95 % of the lanes busy; else the split with the fewest dead
// lanes (groups missing in a row's last slab, threads beyond RL * SG), the widest such slab (longer contiguous pieces of a
// row) that still gives the launch ~500 workgroups — small launches are bound by their latency chain, not by lanes, and
// ran slower with half the workgroups (tools/bench_bn.py: 20 x 20 x 1 392, 19 -> 28 us) — and never pieces under 128 bytes.
static void bn_slab_plan(int C8, int chunks, int* slab_groups, int* nslab) {
  const int old_n = (C8 + 7) / 8;
  auto score = [&](int n) {
    const int g = (C8 + n - 1) / n;
    return (double)C8 * (TR_THREADS / g) / ((double)n * TR_THREADS);
  };
  *slab_groups = C8 < 8 ? C8 : 8;
  *nslab = old_n;
  if (C8 >= 8 && (double)C8 / (8.0 * old_n) >= 0.95) return;   // (the last 64-channel slab is the only partial one)
  double best = 0.0;
  for (int n = 1; n <= old_n; ++n) {
    const int g = (C8 + n - 1) / n;
    if (g > 64 || (g < 8 && n > 1)) continue;
    if (score(n) > best) best = score(n);
  }
  for (int n = 1; n <= old_n; ++n) {
    const int g = (C8 + n - 1) / n;
    if (g > 64 || (g < 8 && n > 1) || score(n) < best - 0.01) continue;
    if ((long long)chunks * n >= 512 || C8 < 8) {
      *slab_groups = g;
      *nslab = n;
      return;
    }
  }
}

// stage 2 of the column reductions: out[which][c] = sum over chunks.  32 channels x 8 chunk lanes
// per workgroup (32 channels x 32 lanes): lane j adds chunks j, j+32, ... in order (double), the 32 lane
// sums are then added in lane order — a fixed association, so the result is deterministic.  (With 8 lanes
// the 32-deep dependent load chain made each of the ~115 launches per step take ~12 us.)
// forward finalize of one channel: sums (local or all-reduced) -> mean, invstd, scale, shift; moving stats
__device__ __forceinline__ void bn_finalize_channel(const BnArgs& a, const BnSegDev& s, int c, double sum, double sumsq) {
  const double n = (double)s.P * (double)a.count_scale;
  const double mean = sum / n;
  double var = sumsq / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float istd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float sc = s.gamma[c] * istd;
  s.fwd[0 * s.C + c] = (float)mean;
  s.fwd[1 * s.C + c] = istd;
  s.fwd[2 * s.C + c] = sc;
  s.fwd[3 * s.C + c] = s.beta[c] - (float)mean * sc;
  if (s.moving_mean) {
    const double vm = a.bessel && n > 1.0 ? var * n / (n - 1.0) : var;
    s.moving_mean[c] = s.moving_mean[c] * a.momentum + (float)mean * (1.0f - a.momentum);
    s.moving_var[c] = s.moving_var[c] * a.momentum + (float)vm * (1.0f - a.momentum);
  }
}

// 1024 threads = CH channels x (1024 / CH) chunk lanes; every lane sums its chunks in double (chunk order), the lane sums
// are added in two fixed levels (16 groups of consecutive lanes, then the 16 group sums in order): deterministic.
//   CH = 16 (64 lanes): short reductions — a lane reads 64 contiguous bytes per chunk;
//   CH = 4 (256 lanes, four times the workgroups): launches with more than 512 chunks on a segment.  The partials of a
//     convolution epilogue are per 32-row block (6 400 on the 80 x 80 head level of the bench batch, 25 600 on a 160 x 160
//     layer): with 64 lanes the 100- to 400-deep chains of dependent loads made 26 of the 115 launches of a step take
//     8 - 33 us (the other 89: ~4 us, the floor of a dependent launch).
// History: 32 channels x 8 lanes ~12 us per launch; 32 x 32 ~9 us.
template <int CH>
__global__ void __launch_bounds__(1024) bn_colreduce_final_kernel(const BnArgs a) {
  constexpr int LANES = 1024 / CH, PER = LANES / 16;
  const BnSegDev& s = a.seg[blockIdx.y];
  __shared__ double red[2][LANES][CH + 1];
  __shared__ double red2[2][16][CH + 1];
  const int cl = threadIdx.x & (CH - 1), lane = threadIdx.x / CH;
  const int c = blockIdx.x * CH + cl;
  double t0 = 0.0, t1 = 0.0;
  if (c < s.C) {
    const float* p = a.ws + a.ws_off[blockIdx.y];
#pragma unroll 8
    for (int k = lane; k < s.chunks; k += LANES) {
      t0 += (double)p[((long long)k * 2 + 0) * s.C + c];
      t1 += (double)p[((long long)k * 2 + 1) * s.C + c];
    }
  }
  red[0][lane][cl] = t0;
  red[1][lane][cl] = t1;
  __syncthreads();
  if (threadIdx.x < 2 * 16 * CH) {   // (which, group of PER consecutive lanes, channel)
    const int which = threadIdx.x / (16 * CH), rem = threadIdx.x - which * (16 * CH);
    const int grp = rem / CH, ch = rem - grp * CH;
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) t += red[which][grp * PER + j][ch];
    red2[which][grp][ch] = t;
  }
  __syncthreads();
  if (threadIdx.x < 2 * CH) {
    const int which = threadIdx.x / CH, ch = threadIdx.x - which * CH;
    const int cc = blockIdx.x * CH + ch;
    if (cc < s.C) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) t += red2[which][j][ch];
      float* out = a.mode == 0 ? s.sums : s.bsums;
      out[which * s.C + cc] = (float)t;
      // gamma / beta gradients are THIS replica's sums (tf.gradients of SyncBatchNormalization: only the
      // optimizer's all-reduce makes them global) — written here, before the caller all-reduces bsums
      if (a.mode == 1) {
        float* gp = which == 0 ? s.dbeta : s.dgamma;
        if (gp) gp[cc] = (float)t;
      }
      red2[which][0][ch] = (double)(float)t;   // for the fused finalize below (same value the unfused path reads)
    }
  }
  if (a.fuse_finalize) {   // single-replica BatchNorm: bn_finalize_kernel's arithmetic without a second launch
    __syncthreads();
    if (threadIdx.x < CH && c < s.C) bn_finalize_channel(a, s, c, red2[0][0][cl], red2[1][0][cl]);
  }
}

// The split form for long reductions.  The partial sums of a convolution epilogue are per 128-row block: 1 600 rows of
// partials on the 80 x 80 head level of the bench batch, 6 400 on a 160 x 160 layer (3 - 13 MB), read ONCE, cold (another
// XCD wrote them) — and a workgroup pulls ~50 GB/s of cold lines, so the C / 16 workgroups of the form above took 11 - 41 us
// on such launches whatever their thread layout (64 or 256 lanes per channel, 4-byte or 16-byte loads: measured,
// tools/probes/bn_final_probe.py) where the reduction of a short segment takes 4.  Here a segment with more than
// BN_KS_CHUNKS partial rows is cut into ks parts of ~256 rows (blockIdx.z): 1024 threads = 4 float4 columns (16 channels)
// x 256 chunk lanes, a wave reads 16 rows x 64 contiguous bytes per load instruction; the 16 lanes of a wave that share a
// column are added by a butterfly (xor 4 .. 32), the 16 wave sums in wave order.  Every part stores its 2 x 16 doubles
// into its slot with write-through (sc1) stores, drains them and takes a ticket with one returning agent-scope atomic
// (the split-K hand-off of the convolution kernels, rn_conv_halo.hip); the part that arrives LAST adds the slots in PART
// order (the same bits whoever is last), zeroes the counter and finishes the channels.  Unsplit segments of the same
// launch (ks = 1) finish directly.
typedef unsigned bn_u32x2_t __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(1024) bn_colreduce_final_split_kernel(const BnArgs a) {
  const BnSegDev& s = a.seg[blockIdx.y];
  const int ks = a.ks[blockIdx.y], part = blockIdx.z;
  if (part >= ks || (int)blockIdx.x * 16 >= s.C) return;
  __shared__ double red[16][2][16];
  __shared__ unsigned s_old;
  const int col = threadIdx.x & 3, lane = threadIdx.x >> 2, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 16 + col * 4;
  double t[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) t[q] = 0.0;
  if (c < s.C) {
    const float* p = a.ws + a.ws_off[blockIdx.y] + c;
    const int per = (s.chunks + ks - 1) / ks;
    const int k1 = (part + 1) * per < s.chunks ? (part + 1) * per : s.chunks;
#pragma unroll 2
    for (int k = part * per + lane; k < k1; k += 256) {
      const float4 v0 = *(const float4*)(p + ((long long)k * 2 + 0) * s.C);
      const float4 v1 = *(const float4*)(p + ((long long)k * 2 + 1) * s.C);
      t[0] += (double)v0.x; t[1] += (double)v0.y; t[2] += (double)v0.z; t[3] += (double)v0.w;
      t[4] += (double)v1.x; t[5] += (double)v1.y; t[6] += (double)v1.z; t[7] += (double)v1.w;
    }
  }
#pragma unroll
  for (int m = 4; m < 64; m <<= 1)
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] += __shfl_xor(t[q], m, 64);
  if ((threadIdx.x & 63) < 4) {
#pragma unroll
    for (int q = 0; q < 8; ++q) red[wave][q >> 2][col * 4 + (q & 3)] = t[q];
  }
  __syncthreads();
  const int which = (threadIdx.x >> 4) & 1, ch = threadIdx.x & 15;   // threads 0..31 finish the 2 x 16 sums
  double r = 0.0;
  if (threadIdx.x < 32) {
#pragma unroll
    for (int j = 0; j < 16; ++j) r += red[j][which][ch];
  }
  if (ks > 1) {
    char* const slots = (char*)a.ws + a.slot_off[blockIdx.y] + (size_t)blockIdx.x * ks * 256;
    if (threadIdx.x < 64) {   // wave 0: 32 stores, drained, one ticket
      if (threadIdx.x < 32) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(slots + part * 256), 0, 256, 0x00020000);
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, r);
        __builtin_amdgcn_raw_buffer_store_b64(bn_u32x2_t{(unsigned)bits, (unsigned)(bits >> 32)}, rs, threadIdx.x * 8, 0, 16);   // sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) {
        unsigned* const cnt = (unsigned*)((char*)a.ws + a.cnt_off[blockIdx.y]) + blockIdx.x;
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(ks - 1)) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
        s_old = old;
      }
    }
    __syncthreads();
    if (s_old != (unsigned)(ks - 1)) return;
    if (threadIdx.x < 32) {   // every part's loads in flight, then the adds in part order; slots past ks read as zeros
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, ks * 256, 0x00020000);
      bn_u32x2_t pv[BN_KS_MAX];
#pragma unroll
      for (int j = 0; j < BN_KS_MAX; ++j) pv[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, j * 256 + threadIdx.x * 8, 0, 16);   // sc1
      r = 0.0;
#pragma unroll
      for (int j = 0; j < BN_KS_MAX; ++j)
        r += __builtin_bit_cast(double, ((unsigned long long)pv[j].y << 32) | pv[j].x);
    }
  }
  if (threadIdx.x < 32) {
    const int cc = blockIdx.x * 16 + ch;
    if (cc < s.C) {
      float* out = a.mode == 0 ? s.sums : s.bsums;
      out[which * s.C + cc] = (float)r;
      if (a.mode == 1) {   // (see bn_colreduce_final_kernel)
        float* gp = which == 0 ? s.dbeta : s.dgamma;
        if (gp) gp[cc] = (float)r;
      }
    }
    red[0][which][ch] = (double)(float)r;
  }
  if (a.fuse_finalize) {
    __syncthreads();
    const int cc = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x < 16 && cc < s.C) bn_finalize_channel(a, s, cc, red[0][0][threadIdx.x], red[0][1][threadIdx.x]);
  }
}

__global__ void bn_finalize_kernel(const BnArgs a) {
  const BnSegDev& s = a.seg[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= s.C) return;
  bn_finalize_channel(a, s, c, (double)s.sums[c], (double)s.sums[s.C + c]);
}

// Which 16-byte units (8 channels of one pixel) a thread of the two elementwise passes walks: first, first + step, ... < last.
//   chunk_u = 0: grid-stride — the stride is the thread count rounded down to a multiple of C/8, so a thread keeps ONE
//     8-channel group for all its rows and holds that group's per-channel parameters in registers;
//   chunk_u = U: workgroup b owns the contiguous units [b * 256 * U, (b + 1) * 256 * U) and a thread takes every 256th of
//     them (same channel group again: the launcher picks this form only when C/8 divides 256).  Not persistent: the
//     hardware hands out the chunks in address order as workgroups retire, and the pass streams at 5.6 - 6.0 TB/s where the
//     grid-stride form of the same loop body reaches 4.9 - 5.5 (tools/probes/stream_probe.hip).
#define BN_ELEMENTWISE_RANGE() BN_ELEMENTWISE_RANGE_(false)
// keep_: threads without work stay (first = last) instead of returning — a workgroup barrier follows the loop
#define BN_ELEMENTWISE_RANGE_(keep_)                                                                  \
  long long first, last, step;                                                                        \
  if (a.chunk_u > 0) {                                                                                \
    const long long span = 256ll * a.chunk_u;                                                         \
    first = blockIdx.x * span + threadIdx.x;                                                          \
    last = (blockIdx.x + 1) * span < total ? (blockIdx.x + 1) * span : total;                         \
    step = 256;                                                                                       \
    if (first >= last) {                                                                              \
      if (!(keep_) || blockIdx.x * span >= total) return;   /* (a whole chunk past the end: uniform) */ \
      first = last;                                                                                   \
    }                                                                                                 \
  } else {                                                                                            \
    const long long nthreads = (long long)gridDim.x * blockDim.x;                                     \
    step = nthreads / C8 * C8;                                                                        \
    first = blockIdx.x * (long long)blockDim.x + threadIdx.x;                                         \
    last = total;                                                                                     \
    if (first >= step) return;                                                                        \
  }

// z = act(y*scale + shift + residual), with a bf16 tensor where the reference has one between two layers: the
// BatchNorm output (when something other than relu / relu6 follows it), the drop_connect output, the residual sum
// in front of tf.nn.swish.  The grid-stride is rounded to a multiple of C/8 so a thread keeps ONE 8-channel group
// for all its rows and holds that group's scale/shift in registers.
__global__ void __launch_bounds__(TR_THREADS) bn_apply_kernel(const BnArgs a) {
  const BnSegDev& s = a.seg[blockIdx.y];
  const int C8 = s.C >> 3;
  const long long total = s.P * C8;
  BN_ELEMENTWISE_RANGE();
  const int c8 = (int)(first % C8);
  float sc[8], sh[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    sc[q] = s.fwd[2 * s.C + c8 * 8 + q];
    sh[q] = s.fwd[3 * s.C + c8 * 8 + q];
  }
  for (long long i = first; i < last; i += step) {
    const bf8 y = unpack8(s.y[i]);
    bf8 o;
    bf8 res;
    if (s.residual) res = unpack8(s.residual[i]);
    const float m = s.sample_scale ? s.sample_scale[(int)(i / C8) / (int)s.rows_per_sample] : 1.0f;
    // one uniform branch per stage, not per element (a runtime test inside the unrolled 8-channel loops compiles
    // to scalar compares and branches PER ELEMENT and makes the pass issue-bound: 40 -> 46 us per launch)
    const bool swish = a.act == RN_ACT_SWISH;
#pragma unroll
    for (int q = 0; q < 8; ++q) o.v[q] = y.v[q] * sc[q] + sh[q];
    if (s.residual != nullptr || s.sample_scale != nullptr || swish) {
#pragma unroll
      for (int q = 0; q < 8; ++q) o.v[q] = rn_rb(o.v[q]);
    }
    if (s.sample_scale) {
#pragma unroll
      for (int q = 0; q < 8; ++q) o.v[q] = rn_rb(o.v[q] * m);
    }
    if (s.residual) {
#pragma unroll
      for (int q = 0; q < 8; ++q) o.v[q] += res.v[q];
      if (swish) {
#pragma unroll
        for (int q = 0; q < 8; ++q) o.v[q] = rn_rb(o.v[q]);
      }
    }
    rn_apply_act_n<8>(o.v, a.act);
    const uint4 zp = pack8(o);
    s.z[i] = zp;
    if (s.mask) {   // gate bits of the STORED bf16 output (what the backward would otherwise re-read as z)
      const bf8 zs = unpack8(zp);
      unsigned bits = 0u;
      // (one uniform branch per vector: act_mask's run-time test inside the unrolled loop was two scalar compares and
      //  branches per ELEMENT in every residual layer of ResNet)
      if (a.act == RN_ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bits |= (zs.v[q] > 0.0f ? 1u : 0u) << q;
      } else if (a.act == RN_ACT_RELU6) {
#pragma unroll
        for (int q = 0; q < 8; ++q) bits |= ((zs.v[q] > 0.0f && zs.v[q] < 6.0f) ? 1u : 0u) << q;
      } else {
        bits = 0xffu;
      }
      s.mask[i] = (unsigned char)bits;
    }
  }
}

// dy = scale*(g - sum_g/n - xhat*sum_gxhat/n); dres (+)= g  (dgamma / dbeta: written by the backward reduction).
// Same fixed-channel-group threading as bn_apply_kernel: 5 per-channel parameters in registers.
template <int G>
__global__ void __launch_bounds__(TR_THREADS) bn_bwd_apply_kernel(const BnArgs a) {
  const BnSegDev& s = a.seg[blockIdx.y];
  const int C8 = s.C >> 3;
  const long long total = s.P * C8;
  const float inv_n = (float)(1.0 / ((double)s.P * (double)a.count_scale));
  // rn_bn_segment.dy_colsum_partial (chunked form only: the host checks): column sums of the dy this workgroup STORES
  const bool colsum = s.colsum != nullptr;
  BN_ELEMENTWISE_RANGE_(colsum);
  const int c8 = a.chunk_u > 0 ? (int)(threadIdx.x % C8) : (int)(first % C8);   // (the same value; valid for first == last too)
  float mean[8], istd[8], sc[8], shq[8], k1[8], k2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = c8 * 8 + q;
    mean[q] = s.fwd[0 * s.C + c];
    istd[q] = s.fwd[1 * s.C + c];
    sc[q] = s.fwd[2 * s.C + c];
    shq[q] = s.fwd[3 * s.C + c];
    k1[q] = s.bsums[c] * inv_n;
    k2[q] = s.bsums[s.C + c] * inv_n;
  }
  float cs[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) cs[q] = 0.0f;
  for (long long i = first; i < last; i += step) {
    const bf8 y = unpack8(s.y[i]);
    const bf8 dz = unpack8(s.dz[i]);
    const bool from_u = !s.residual && (a.act == RN_ACT_RELU || a.act == RN_ACT_RELU6);
    constexpr bool need_z = G == G_Z_RELU || G == G_Z_RELU6 || G == G_GENERIC;
    bf8 z;
    if (need_z && (G != G_GENERIC || (a.act != RN_ACT_NONE && !from_u))) z = unpack8(s.z[i]);
    const float m = s.sample_scale ? s.sample_scale[(int)(i / C8) / (int)s.rows_per_sample] : 1.0f;
    unsigned bits = 0xffu;
    if (G == G_MASK) bits = s.mask[i];
    bf8 g, o;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float u = y.v[q] * sc[q] + shq[q];
      g.v[q] = grad_gate<G>(dz.v[q], z.v[q], u, a.act, from_u);
      if (G == G_MASK) g.v[q] = ((bits >> q) & 1u) ? g.v[q] : 0.0f;
      const float xh = (y.v[q] - mean[q]) * istd[q];
      o.v[q] = sc[q] * (g.v[q] * m - k1[q] - xh * k2[q]);
    }
    const uint4 dyp = pack8(o);
    s.dy[i] = dyp;
    if (colsum) {
      const bf8 st = unpack8(dyp);
#pragma unroll
      for (int q = 0; q < 8; ++q) cs[q] += st.v[q];
    }
    if (s.dres) {
      if (s.dres_accumulate) {
        const bf8 old = unpack8(s.dres[i]);
#pragma unroll
        for (int q = 0; q < 8; ++q) g.v[q] += old.v[q];
      }
      s.dres[i] = pack8(g);
    }
  }
  if (colsum) {
    // chunked form: thread t of every chunk holds channel group t % C8 (C8 divides 256 and the chunk's first unit is a
    // multiple of 256); the 256 / C8 threads of a group are added in thread order: deterministic
    __shared__ float red[TR_THREADS][9];
#pragma unroll
    for (int q = 0; q < 8; ++q) red[threadIdx.x][q] = cs[q];
    __syncthreads();
    if ((int)threadIdx.x < C8) {
      float t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = 0.0f;
      for (int j = threadIdx.x; j < TR_THREADS; j += C8)
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] += red[j][q];
      float* dst = s.colsum + ((long long)blockIdx.x * 2) * s.C + threadIdx.x * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[q] = t[q];
    }
  }
}

// row chunks of the stage-1 partial sums of a segment: ~2048 workgroups per segment over (row chunks x 64-channel
// slabs) — narrow layers (EfficientNet: 24..144 channels = 1..3 slabs) get more row chunks, so the reduction still fills
// the chip — or what the producing convolution wrote (rn_bn_segment.ext_chunks / ext_chunks_bwd)
static int bn_chunks_of(const rn_bn_segment& s, int mode, int* rows_per_chunk) {
  const long long want_chunks = 2048 / rn_cdiv(s.C, 64) < 256 ? 256 : 2048 / rn_cdiv(s.C, 64);
  long long rpc = rn_cdiv(rn_cdiv(s.P, want_chunks), 32) * 32;  // multiple of 32 rows
  if (rpc < 32) rpc = 32;
  if (rows_per_chunk) *rows_per_chunk = (int)rpc;
  int chunks = (int)rn_cdiv(s.P, rpc);
  if (mode == 0 && s.ext_chunks > 0) chunks = s.ext_chunks;
  if (mode == 1 && s.ext_chunks_bwd > 0) chunks = s.ext_chunks_bwd;
  return chunks;
}

// mode 0 (forward statistics) honours rn_bn_segment.ext_chunks: stage-1 partials written by the conv epilogue
static int bn_fill(const rn_bn_problem* p, BnArgs& a, int need_ws, int mode = 1) {
  if (!p || p->num_segments < 1 || p->num_segments > RN_CONV_MAX_SEGMENTS) return -1;
  a.nseg = p->num_segments; a.act = p->act; a.bessel = p->bessel; a.mode = 0; a.fuse_finalize = 0; a.chunk_u = 0;
  a.eps = p->eps; a.momentum = p->momentum; a.count_scale = p->count_scale > 0 ? p->count_scale : 1.0f;
  long long off = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_bn_segment& s = p->seg[i];
    if (s.P <= 0 || s.C <= 0 || (s.C % 8)) return -1;
    BnSegDev& d = a.seg[i];
    d.y = (const uint4*)s.y; d.z = (uint4*)s.z; d.residual = (const uint4*)s.residual;
    d.dz = (const uint4*)s.dz; d.dy = (uint4*)s.dy; d.dres = (uint4*)s.dres;
    d.sums = s.sums; d.fwd = s.fwd; d.bsums = s.bsums; d.gamma = s.gamma; d.beta = s.beta;
    d.moving_mean = s.moving_mean; d.moving_var = s.moving_var; d.dgamma = s.dgamma; d.dbeta = s.dbeta;
    d.P = s.P; d.C = s.C; d.dres_accumulate = s.dres_accumulate;
    d.sample_scale = s.sample_scale; d.rows_per_sample = s.rows_per_sample > 0 ? s.rows_per_sample : 1;
    d.mask = (unsigned char*)s.act_mask;
    d.colsum = s.dy_colsum_partial;
    if (s.sample_scale && p->act == RN_ACT_SWISH) return -1;   // swish' is recomputed without the factor
    d.chunks = bn_chunks_of(s, mode, &d.rows_per_chunk);
    bn_slab_plan(s.C / 8, d.chunks, &d.slab_groups, &d.nslab);
    a.ws_off[i] = off;
    off += (long long)d.chunks * 2 * s.C;
  }
  // Workspace tail, behind the partials of WHICHEVER mode needs more of them (the ticket counters must never be written
  // by a partial sum): one counter per (segment, 16-channel block), then the part slots of the split segments.
  long long other = 0;
  for (int i = 0; i < p->num_segments; ++i) other += (long long)bn_chunks_of(p->seg[i], 1 - mode, nullptr) * 2 * p->seg[i].C;
  long long tail = ((off > other ? off : other) * 4 + 255) / 256 * 256;
  for (int i = 0; i < p->num_segments; ++i) {
    a.cnt_off[i] = tail;
    tail += rn_cdiv(p->seg[i].C, 16) * 4;
  }
  tail = (tail + 255) / 256 * 256;
  for (int i = 0; i < p->num_segments; ++i) {
    const int ch = a.seg[i].chunks;
    int ks = ch > BN_KS_CHUNKS ? (int)rn_cdiv(ch, 256) : 1;
    if (ks > BN_KS_MAX) ks = BN_KS_MAX;
    a.ks[i] = ks;
    a.slot_off[i] = tail;
    if (ks > 1) tail += rn_cdiv(p->seg[i].C, 16) * ks * 256;   // 2 x 16 doubles per (block, part)
  }
  a.total_bytes = tail;
  (void)need_ws;
  return 0;
}

// [partial sums of the mode that needs more][ticket counters][part slots].  The ticket counters must be ZERO before the
// first launch on a workspace (zero the workspace once after allocating it); every launch leaves them zero.
extern "C" size_t rn_bn_workspace_bytes(const rn_bn_problem* p) {
  size_t need = 0;
  for (int mode = 0; mode < 2; ++mode) {     // forward (possibly external partials) and backward chunking
    BnArgs a;
    if (bn_fill(p, a, 1, mode)) return 0;
    if ((size_t)a.total_bytes > need) need = (size_t)a.total_bytes;
  }
  return need;
}

// Zeroes the ticket counters (and nothing else) of a workspace allocated for `p`: the cheap way to honour the "counters
// start at zero" contract for a workspace that came out of an uninitialised allocation.  The counter region is
// the same for both modes (bn_fill places it behind the larger of the two partial areas).
extern "C" int rn_bn_workspace_init(const rn_bn_problem* p, void* ws, size_t ws_bytes, void* stream) {
  BnArgs a;
  RN_CHECK_ARG(p && bn_fill(p, a, 1, 0) == 0, "rn_bn_workspace_init: bad problem (C %% 8 == 0, 1..10 segments)");
  const size_t need = rn_bn_workspace_bytes(p);
  if (!ws || ws_bytes < need) {
    rn_set_error("rn_bn_workspace_init: workspace too small (%zu < %zu)", ws_bytes, need);
    return RN_ENOMEM;
  }
  const long long begin = a.cnt_off[0];
  long long end = begin;
  for (int i = 0; i < p->num_segments; ++i) end = a.cnt_off[i] + rn_cdiv(p->seg[i].C, 16) * 4;
  RN_CHECK_HIP(hipMemsetAsync((char*)ws + begin, 0, (size_t)(end - begin), (hipStream_t)stream));
  return RN_OK;
}

extern "C" size_t rn_bn_partial_offset_bytes(const rn_bn_problem* p, int segment) {
  BnArgs a;
  if (bn_fill(p, a, 1, 0) || segment < 0 || segment >= a.nseg) return 0;
  return (size_t)a.ws_off[segment] * sizeof(float);
}

extern "C" size_t rn_bn_bwd_partial_offset_bytes(const rn_bn_problem* p, int segment) {
  BnArgs a;
  if (bn_fill(p, a, 1, 1) || segment < 0 || segment >= a.nseg) return 0;
  return (size_t)a.ws_off[segment] * sizeof(float);
}

static int bn_colreduce(const rn_bn_problem* p, int mode, void* ws, size_t ws_bytes, hipStream_t st,
                        const char* fn, int fuse_finalize = 0) {
  BnArgs a;
  RN_CHECK_ARG(bn_fill(p, a, 1, mode) == 0, "%s: bad problem (C %% 8 == 0, 1..10 segments)", fn);
  int n_ext = 0;
  for (int i = 0; i < p->num_segments; ++i)
    n_ext += ((mode == 0 && p->seg[i].ext_chunks > 0) || (mode == 1 && p->seg[i].ext_chunks_bwd > 0)) ? 1 : 0;
  RN_CHECK_ARG(n_ext == 0 || n_ext == p->num_segments, "%s: ext_chunks must be set on all segments or none", fn);
  if (mode == 1 && n_ext) {
    // what the data-gradient epilogue can reproduce: the u > 0 mask of a ReLU without residual input (or no
    // activation at all: scale = 0, shift = 1 is the caller's business then), no per-sample factors
    RN_CHECK_ARG(bn_gate_mode(p) == G_U_RELU, "%s: ext_chunks_bwd needs act = relu without residual inputs", fn);
    for (int i = 0; i < p->num_segments; ++i)
      RN_CHECK_ARG(!p->seg[i].sample_scale, "%s: ext_chunks_bwd with sample_scale", fn);
  }
  if (!ws || ws_bytes < rn_bn_workspace_bytes(p)) {
    rn_set_error("%s: workspace too small", fn);
    return RN_ENOMEM;
  }
  a.mode = mode;
  a.fuse_finalize = fuse_finalize;
  if (fuse_finalize)
    for (int i = 0; i < a.nseg; ++i)
      RN_CHECK_ARG(a.seg[i].fwd && a.seg[i].gamma && a.seg[i].beta, "%s: null tensor", fn);
  a.ws = (float*)ws;
  int max_chunks = 0, max_slabs = 0, max_c = 0;
  for (int i = 0; i < a.nseg; ++i) {
    const BnSegDev& s = a.seg[i];
    RN_CHECK_ARG(s.y && (mode == 0 ? s.sums != nullptr : (s.dz && s.bsums && s.fwd)), "%s: null tensor", fn);
    RN_CHECK_ARG(mode == 0 || a.act == RN_ACT_NONE || s.z || s.mask, "%s: z needed for the activation mask", fn);
    if (s.chunks > max_chunks) max_chunks = s.chunks;
    if (s.nslab > max_slabs) max_slabs = s.nslab;
    if (s.C > max_c) max_c = s.C;
  }
  if (n_ext == 0) {   // otherwise the producing convolution already wrote the stage-1 partials
    const dim3 grid(max_chunks, max_slabs, a.nseg), block(TR_THREADS);
    bool all8 = true;
    for (int i = 0; i < a.nseg; ++i) all8 = all8 && a.seg[i].slab_groups == 8;
    if (mode == 0) {
      if (all8) hipLaunchKernelGGL((bn_colreduce_kernel<-1, 8>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((bn_colreduce_kernel<-1, 0>), grid, block, 0, st, a);
    } else {
#define BN_CALL_(G_)                                                                       \
  if (all8) hipLaunchKernelGGL((bn_colreduce_kernel<G_, 8>), grid, block, 0, st, a);       \
  else hipLaunchKernelGGL((bn_colreduce_kernel<G_, 0>), grid, block, 0, st, a)
      BN_DISPATCH_GATE(bn_gate_mode(p), BN_CALL_)
#undef BN_CALL_
    }
    RN_CHECK_LAUNCH();
  }
  int max_ks = 1;
  for (int i = 0; i < a.nseg; ++i) max_ks = a.ks[i] > max_ks ? a.ks[i] : max_ks;
  static const int final_form = getenv("RNET_BN_FINAL") ? atoi(getenv("RNET_BN_FINAL")) : 0;   // A/B probe: 1 = never split
  if (max_ks > 1 && final_form != 1 && (((uintptr_t)a.ws) & 15) == 0)
    hipLaunchKernelGGL(bn_colreduce_final_split_kernel, dim3((max_c + 15) / 16, a.nseg, max_ks), dim3(1024), 0, st, a);
  else
    hipLaunchKernelGGL(bn_colreduce_final_kernel<16>, dim3((max_c + 15) / 16, a.nseg), dim3(1024), 0, st, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_bn_stats(const rn_bn_problem* p, void* ws, size_t ws_bytes, void* stream) {
  return bn_colreduce(p, 0, ws, ws_bytes, (hipStream_t)stream, "rn_bn_stats");
}
extern "C" int rn_bn_stats_finalize(const rn_bn_problem* p, void* ws, size_t ws_bytes, void* stream) {
  return bn_colreduce(p, 0, ws, ws_bytes, (hipStream_t)stream, "rn_bn_stats_finalize", 1);
}
extern "C" int rn_bn_bwd_reduce(const rn_bn_problem* p, void* ws, size_t ws_bytes, void* stream) {
  return bn_colreduce(p, 1, ws, ws_bytes, (hipStream_t)stream, "rn_bn_bwd_reduce");
}

extern "C" int rn_bn_finalize(const rn_bn_problem* p, void* stream) {
  BnArgs a;
  RN_CHECK_ARG(bn_fill(p, a, 0) == 0, "rn_bn_finalize: bad problem");
  int max_c = 0;
  for (int i = 0; i < a.nseg; ++i) {
    RN_CHECK_ARG(a.seg[i].sums && a.seg[i].fwd && a.seg[i].gamma && a.seg[i].beta, "rn_bn_finalize: null tensor");
    if (a.seg[i].C > max_c) max_c = a.seg[i].C;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((max_c + 255) / 256, a.nseg), dim3(256), 0, (hipStream_t)stream, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// grid of the elementwise passes; picks the chunked form (BN_ELEMENTWISE_RANGE) when every segment's C/8 divides 256.
// U = 8 units per thread: each thread re-reads its 16 - 48 per-channel parameters per chunk, which is what U amortises
// (training step on one box, tools/probes/ab_bn_chunk.sh: grid-stride 31.29 ms, U = 1: 33.1, 2: 31.9, 4: 31.14, 8: 31.09,
// 16: 31.38, 32: 32.10 — few, long workgroups lose the ordered hand-out again).
#ifndef RN_BN_CHUNK_U
#define RN_BN_CHUNK_U 8
#endif
static int bn_elementwise_grid(BnArgs& a, long long max_units) {
  static_assert(TR_THREADS == 256, "BN_ELEMENTWISE_RANGE assumes 256 threads");
  bool chunk = RN_BN_CHUNK_U > 0;
  for (int i = 0; i < a.nseg; ++i) chunk = chunk && 256 % (a.seg[i].C >> 3) == 0;
  const long long blocks = rn_cdiv(max_units, 256ll * (RN_BN_CHUNK_U > 0 ? RN_BN_CHUNK_U : 1));
  if (chunk && blocks < (1ll << 31)) {
    a.chunk_u = RN_BN_CHUNK_U;
    return (int)blocks;
  }
  a.chunk_u = 0;
  return tr_blocks(max_units, 4096);
}

extern "C" int rn_bn_apply(const rn_bn_problem* p, void* stream) {
  BnArgs a;
  RN_CHECK_ARG(bn_fill(p, a, 0) == 0, "rn_bn_apply: bad problem");
  long long mx = 0;
  for (int i = 0; i < a.nseg; ++i) {
    RN_CHECK_ARG(a.seg[i].y && a.seg[i].z && a.seg[i].fwd, "rn_bn_apply: null tensor");
    const long long t = a.seg[i].P * (a.seg[i].C / 8);
    if (t > mx) mx = t;
  }
  hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_elementwise_grid(a, mx), a.nseg), dim3(TR_THREADS), 0, (hipStream_t)stream, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_bn_bwd_colsum_chunks(const rn_bn_problem* p, int segment) {
  BnArgs a;
  if (bn_fill(p, a, 0) || segment < 0 || segment >= a.nseg) return 0;
  long long mx = 0;
  for (int i = 0; i < a.nseg; ++i) {
    const long long t = a.seg[i].P * (a.seg[i].C / 8);
    if (t > mx) mx = t;
  }
  bn_elementwise_grid(a, mx);
  if (a.chunk_u <= 0) return 0;
  return (int)rn_cdiv(a.seg[segment].P * (a.seg[segment].C / 8), 256ll * a.chunk_u);
}

extern "C" int rn_bn_bwd_apply(const rn_bn_problem* p, void* stream) {
  BnArgs a;
  RN_CHECK_ARG(bn_fill(p, a, 0) == 0, "rn_bn_bwd_apply: bad problem");
  long long mx = 0;
  for (int i = 0; i < a.nseg; ++i) {
    RN_CHECK_ARG(a.seg[i].y && a.seg[i].dz && a.seg[i].dy && a.seg[i].fwd && a.seg[i].bsums,
                 "rn_bn_bwd_apply: null tensor");
    RN_CHECK_ARG(a.act == RN_ACT_NONE || a.seg[i].z || a.seg[i].mask, "rn_bn_bwd_apply: z needed for the activation mask");
    const long long t = a.seg[i].P * (a.seg[i].C / 8);
    if (t > mx) mx = t;
  }
  {
    const dim3 grid(bn_elementwise_grid(a, mx), a.nseg), block(TR_THREADS);
    for (int i = 0; i < a.nseg; ++i)
      RN_CHECK_ARG(!a.seg[i].colsum || a.chunk_u > 0,
                   "rn_bn_bwd_apply: dy_colsum_partial needs the chunked form (rn_bn_bwd_colsum_chunks() > 0)");
#define BN_CALL_(G_) hipLaunchKernelGGL(bn_bwd_apply_kernel<G_>, grid, block, 0, (hipStream_t)stream, a)
    BN_DISPATCH_GATE(bn_gate_mode(p), BN_CALL_)
#undef BN_CALL_
  }
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- gradient of an activation without BN (prediction-free layers): dy = dz * mask(z) --------
__global__ void __launch_bounds__(TR_THREADS)
act_bwd_kernel(const uint4* __restrict__ dz, const uint4* __restrict__ z, uint4* __restrict__ dy, long long n8,
               int act) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8;
       i += (long long)gridDim.x * blockDim.x) {
    bf8 g = unpack8(dz[i]);
    const bf8 zz = unpack8(z[i]);
#pragma unroll
    for (int q = 0; q < 8; ++q) g.v[q] *= act_mask(zz.v[q], act);
    dy[i] = pack8(g);
  }
}

// ---- f32 -> bf16 cast (loss gradients of the fp32 prediction convs feed bf16 dgrad/wgrad) ------
__global__ void __launch_bounds__(TR_THREADS)
cast_f32_bf16_kernel(const float4* __restrict__ x, uint2* __restrict__ y, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    uint2 o;
    o.x = rn_pack_bf16x2(v.x, v.y);
    o.y = rn_pack_bf16x2(v.z, v.w);
    y[i] = o;
  }
}
__global__ void __launch_bounds__(TR_THREADS)
cast_pad_kernel(const float* __restrict__ x, uint2* __restrict__ y, long long P, int C, int Cpad) {
  const int q = Cpad / 4;
  const long long total = P * q;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / q;
    const int c = (int)(i - r * q) * 4;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (c + u) < C ? x[r * C + c + u] : 0.0f;
    uint2 o;
    o.x = rn_pack_bf16x2(v[0], v[1]);
    o.y = rn_pack_bf16x2(v[2], v[3]);
    y[i] = o;
  }
}
extern "C" int rn_cast_pad_f32_to_bf16(const float* x, void* y, int64_t P, int C, int Cpad, void* stream) {
  RN_CHECK_ARG(x && y && P > 0 && C > 0 && Cpad >= C && Cpad % 4 == 0, "rn_cast_pad_f32_to_bf16: bad argument");
  hipLaunchKernelGGL(cast_pad_kernel, dim3(tr_blocks(P * (Cpad / 4))), dim3(TR_THREADS), 0, (hipStream_t)stream, x,
                     (uint2*)y, (long long)P, C, Cpad);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
extern "C" int rn_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  RN_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "rn_cast_f32_to_bf16: bad argument (n %% 4 == 0)");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(tr_blocks(n / 4)), dim3(TR_THREADS), 0, (hipStream_t)stream,
                     (const float4*)x, (uint2*)y, (long long)(n / 4));
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- dst[c] = add + sum over rows of src[r * row_stride + c], rows added in index order (deterministic) -----------------
// The bias gradient of a conv shared by the pyramid levels = the sum of its per-level column sums (a handful of rows), and
// the per-image positive counts of a batch: one thread per column, no library reduction on the product path.
__global__ void __launch_bounds__(TR_THREADS)
reduce_rows_kernel(const float* __restrict__ src, int rows, long long row_stride, int n, float add, float* __restrict__ dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s = 0.0f;
  for (int r = 0; r < rows; ++r) s += src[r * row_stride + c];
  dst[c] = s + add;
}
extern "C" int rn_reduce_rows_f32(const float* src, int rows, int64_t row_stride, int n, float add, float* dst, void* stream) {
  RN_CHECK_ARG(src && dst && rows > 0 && n > 0 && row_stride >= 0, "rn_reduce_rows_f32: bad argument");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((n + TR_THREADS - 1) / TR_THREADS), dim3(TR_THREADS), 0, (hipStream_t)stream,
                     src, rows, (long long)row_stride, n, add, dst);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- zero-insertion upsample: y[n,2h,2w,:] = x[n,h,w,:], zeros elsewhere (dgrad of stride 2) ---
__global__ void __launch_bounds__(TR_THREADS)
upsample_zero_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8, int Ho,
                     int Wo) {
  const long long total = (long long)N * Ho * Wo * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const RnIdx4 d_ = rn_decode4(i, C8, Wo, Ho, rn_decode_mode(total, C8));
    const int c = d_.c, ox = d_.x, oy = d_.y, n = d_.n;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!(ox & 1) && !(oy & 1) && (oy >> 1) < H && (ox >> 1) < W)
      v = x[(((long long)n * H + (oy >> 1)) * W + (ox >> 1)) * C8 + c];
    y[i] = v;
  }
}
extern "C" int rn_upsample_zero2x(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo,
                                  void* stream) {
  RN_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C % 8 == 0 && Ho >= 2 * H - 1 && Wo >= 2 * W - 1,
               "rn_upsample_zero2x: bad argument");
  hipLaunchKernelGGL(upsample_zero_kernel, dim3(tr_blocks((long long)N * Ho * Wo * (C / 8))), dim3(TR_THREADS), 0,
                     (hipStream_t)stream, (const uint4*)x, (uint4*)y, N, H, W, C / 8, Ho, Wo);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- y[n,2h,2w,:] += x[n,h,w,:] (fp32 add, one rounding): the data gradient of a 1x1 / stride-2 convolution
// only has values at the even positions, so its GEMM runs at the low resolution and this scatters the result
__global__ void __launch_bounds__(TR_THREADS)
scatter_add2x_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8, int Ho, int Wo) {
  const long long total = (long long)N * H * W * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const RnIdx4 d_ = rn_decode4(i, C8, W, H, rn_decode_mode(total, C8));
    const int c = d_.c, w = d_.x, h = d_.y, n = d_.n;
    if (2 * h >= Ho || 2 * w >= Wo) continue;
    const long long o = (((long long)n * Ho + 2 * h) * Wo + 2 * w) * C8 + c;
    const bf8 a = unpack8(x[i]), b = unpack8(y[o]);
    bf8 r;
#pragma unroll
    for (int q = 0; q < 8; ++q) r.v[q] = a.v[q] + b.v[q];
    y[o] = pack8(r);
  }
}
extern "C" int rn_scatter_add2x(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int accumulate,
                                void* stream) {
  RN_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C % 8 == 0 && Ho >= 2 * H - 1 && Wo >= 2 * W - 1,
               "rn_scatter_add2x: bad argument");
  if (!accumulate) return rn_upsample_zero2x(x, y, N, H, W, C, Ho, Wo, stream);   // first writer: zeros elsewhere
  hipLaunchKernelGGL(scatter_add2x_kernel, dim3(tr_blocks((long long)N * H * W * (C / 8))), dim3(TR_THREADS), 0,
                     (hipStream_t)stream, (const uint4*)x, (uint4*)y, N, H, W, C / 8, Ho, Wo);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// depth-to-space by 2: y[n, 2i + a, 2j + b, c] (+)= x[n, i, j, (a*2 + b)*C + c] — puts the four phases of the
// sub-pixel stride-2 data gradient (rn_dgrad_pack.pad_ == 1) in place; accumulate adds in fp32, one rounding
__global__ void __launch_bounds__(TR_THREADS)
depth_to_space2x_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W, int C8, int accumulate) {
  const long long total = (long long)N * H * W * 4 * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // [N][H][W][4 phases][C8] = an [N][H][4 W][C8] tensor whose column is 4 w + phase
    const RnIdx4 d_ = rn_decode4(i, C8, 4 * W, H, rn_decode_mode(total, C8));
    const int c = d_.c, ph = d_.x & 3, w = d_.x >> 2, h = d_.y, n = d_.n;
    const long long o = (((long long)n * 2 * H + 2 * h + (ph >> 1)) * 2 * W + 2 * w + (ph & 1)) * C8 + c;
    uint4 v = x[i];
    if (accumulate) {
      const bf8 a = unpack8(v), b = unpack8(y[o]);
      bf8 r;
#pragma unroll
      for (int q = 0; q < 8; ++q) r.v[q] = a.v[q] + b.v[q];
      v = pack8(r);
    }
    y[o] = v;
  }
}
extern "C" int rn_depth_to_space2x(const void* x, void* y, int N, int H, int W, int C, int accumulate, void* stream) {
  RN_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "rn_depth_to_space2x: bad argument");
  hipLaunchKernelGGL(depth_to_space2x_kernel, dim3(tr_blocks((long long)N * H * W * 4 * (C / 8))), dim3(TR_THREADS), 0,
                     (hipStream_t)stream, (const uint4*)x, (uint4*)y, N, H, W, C / 8, accumulate);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_act_bwd(const void* dz, const void* z, void* dy, int64_t n, int act, void* stream) {
  RN_CHECK_ARG(dz && z && dy && n > 0 && n % 8 == 0, "rn_act_bwd: bad argument");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(tr_blocks(n / 8)), dim3(TR_THREADS), 0, (hipStream_t)stream,
                     (const uint4*)dz, (const uint4*)z, (uint4*)dy, (long long)(n / 8), act);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- max-pool backward (gather form, any k / stride / TF-SAME pads): every INPUT pixel sums dy of
// the windows that contain it and whose FIRST maximum (row-major scan, padded taps skipped) it is.
// No atomics, deterministic; <= ceil(k/stride)^2 windows x k^2 taps of L2-resident reads per pixel.
__global__ void __launch_bounds__(TR_THREADS)
maxpool_bwd_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy, uint4* __restrict__ dx, int N, int H,
                   int W, int C8, int k, int stride, int pt, int pl, int Ho, int Wo, int accumulate) {
  const long long total = (long long)N * H * W * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const RnIdx4 d_ = rn_decode4(i, C8, W, H, rn_decode_mode(total, C8));
    const int c = d_.c, ix = d_.x, iy = d_.y, n = d_.n;
    const bf8 me = unpack8(x[i]);
    bf8 g;
    if (accumulate) g = unpack8(dx[i]);
    else {
#pragma unroll
      for (int q = 0; q < 8; ++q) g.v[q] = 0.0f;
    }
    // windows oy with oy*stride - pt <= iy <= oy*stride - pt + k - 1
    int oy0 = (iy + pt - (k - 1) + stride - 1) / stride;
    if (iy + pt - (k - 1) < 0) oy0 = 0;
    int ox0 = (ix + pl - (k - 1) + stride - 1) / stride;
    if (ix + pl - (k - 1) < 0) ox0 = 0;
    const int oy1 = (iy + pt) / stride, ox1 = (ix + pl) / stride;
    for (int oy = oy0; oy <= oy1 && oy < Ho; ++oy)
      for (int ox = ox0; ox <= ox1 && ox < Wo; ++ox) {
        const int my_r = iy - (oy * stride - pt), my_s = ix - (ox * stride - pl);
        bool win[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) win[q] = true;
        for (int r = 0; r < k; ++r) {
          const int yy = oy * stride - pt + r;
          if ((unsigned)yy >= (unsigned)H) continue;
          for (int s = 0; s < k; ++s) {
            const int xx = ox * stride - pl + s;
            if ((unsigned)xx >= (unsigned)W || (r == my_r && s == my_s)) continue;
            const bf8 v = unpack8(x[(((long long)n * H + yy) * W + xx) * C8 + c]);
            const bool before = r < my_r || (r == my_r && s < my_s);
#pragma unroll
            for (int q = 0; q < 8; ++q)   // an earlier tap wins ties, a later one must be strictly larger
              if (before ? (v.v[q] >= me.v[q]) : (v.v[q] > me.v[q])) win[q] = false;
          }
        }
        const bf8 d = unpack8(dy[(((long long)n * Ho + oy) * Wo + ox) * C8 + c]);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (win[q]) g.v[q] += d.v[q];
      }
    dx[i] = pack8(g);
  }
}
extern "C" int rn_maxpool2d_nhwc_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int k,
                                     int stride, int pad_top, int pad_left, int Ho, int Wo, int accumulate,
                                     void* stream) {
  RN_CHECK_ARG(x && dy && dx && C % 8 == 0 && k >= 1 && stride >= 1 && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0,
               "rn_maxpool2d_nhwc_bwd: bad argument");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(tr_blocks((long long)N * H * W * (C / 8))), dim3(TR_THREADS), 0,
                     (hipStream_t)stream, (const uint4*)x, (const uint4*)dy, (uint4*)dx, N, H, W, C / 8, k, stride,
                     pad_top, pad_left, Ho, Wo, accumulate);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- FPN top-down backward, one level: din = (dout + sum2x2(din_finer)) * mask(out) ------------
__global__ void __launch_bounds__(TR_THREADS)
topdown_bwd_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ din_finer,
                   const uint4* __restrict__ out, uint4* __restrict__ din, int N, int H, int W, int C8, int act) {
  const long long total = (long long)N * H * W * C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const RnIdx4 d_ = rn_decode4(i, C8, W, H, rn_decode_mode(total, C8));
    const int c = d_.c, x = d_.x, y = d_.y, n = d_.n;
    bf8 g = unpack8(dout[i]);
    if (din_finer) {
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          const bf8 v = unpack8(din_finer[(((long long)n * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * C8 + c]);
#pragma unroll
          for (int q = 0; q < 8; ++q) g.v[q] += v.v[q];
        }
    }
    if (out && act != RN_ACT_NONE) {
      const bf8 z = unpack8(out[i]);
#pragma unroll
      for (int q = 0; q < 8; ++q) g.v[q] *= act_mask(z.v[q], act);
    }
    din[i] = pack8(g);
  }
}
extern "C" int rn_fpn_topdown_bwd_level(const void* dout, const void* din_finer, const void* out, void* din, int N,
                                        int H, int W, int C, int act, void* stream) {
  RN_CHECK_ARG(dout && din && C % 8 == 0 && N > 0 && H > 0 && W > 0, "rn_fpn_topdown_bwd_level: bad argument");
  hipLaunchKernelGGL(topdown_bwd_kernel, dim3(tr_blocks((long long)N * H * W * (C / 8))), dim3(TR_THREADS), 0,
                     (hipStream_t)stream, (const uint4*)dout, (const uint4*)din_finer, (const uint4*)out,
                     (uint4*)din, N, H, W, C / 8, act);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- BalanceFeatures backward (balance_features.py:19-60) -----------------------------------
#define RN_PYR_MAX 8
// (n, y, x, channel group) of a flat element index with 32-bit divisions when the tensor has fewer than 2^31 elements:
// six 64-bit divisions by run-time values per thread made balance_bwd_in_kernel VALU-bound (8.7 M threads at B = 32)
typedef RnIdx4 TrIdx4;   // rn_common.h: rn_decode4
__device__ __forceinline__ TrIdx4 tr_decode4(long long t, int C8, int Wl, int Hl, int mode) { return rn_decode4(t, C8, Wl, Hl, mode); }

struct BalBwd {
  int L, mid, N, H0, W0, C8;
  const uint4* dout[RN_PYR_MAX];
  const uint4* in[RN_PYR_MAX];  // forward inputs (only levels < mid are read: max-pool argmax)
  uint4* din[RN_PYR_MAX];
  const uint4* avg;             // forward average at the intermediate level
  uint4* davg;                  // scratch
  uint2* arg[RN_PYR_MAX];       // levels > mid: per (coarse pixel, 8-channel group) the window position (dy * f + dx, one byte per
                                // channel) of the first maximum of avg — written once by balance_bwd_arg_kernel
  long long begin[RN_PYR_MAX + 1];
};

// Levels coarser than the intermediate one receive max_pool(avg): their gradient goes to the first maximum of avg in the
// f x f window.  Every fine pixel of a window used to recompute that argmax for itself (64 reads per pixel and level for the
// coarsest level: 92 x 16 bytes per thread, 184 us per step at B = 32); now one thread per COARSE pixel finds it once
// (loads eight at a time) and the fine pixels read one byte per channel.
__global__ void __launch_bounds__(TR_THREADS) balance_bwd_arg_kernel(BalBwd b) {
  const int Hm = b.H0 >> b.mid, Wm = b.W0 >> b.mid;
  long long total = 0;
  for (int l = b.mid + 1; l < b.L; ++l) total += (long long)b.N * (b.H0 >> l) * (b.W0 >> l) * b.C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = b.mid + 1;
    long long t = i;
    while (true) {
      const long long nl = (long long)b.N * (b.H0 >> l) * (b.W0 >> l) * b.C8;
      if (t < nl) break;
      t -= nl;
      ++l;
    }
    const int Hl = b.H0 >> l, Wl = b.W0 >> l, f = 1 << (l - b.mid);
    const unsigned u0 = (unsigned)t;
    const int c = (int)(u0 % (unsigned)b.C8);
    unsigned u = u0 / (unsigned)b.C8;
    const int wx = (int)(u % (unsigned)Wl);
    u /= (unsigned)Wl;
    const int wy = (int)(u % (unsigned)Hl), n = (int)(u / (unsigned)Hl);
    float best[8];
    int arg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { best[q] = -INFINITY; arg[q] = 0; }
    const uint4* row0 = b.avg + (((long long)n * Hm) + (long long)wy * f) * Wm * b.C8 + (long long)(wx * f) * b.C8 + c;
    for (int dy = 0; dy < f; ++dy) {
      const uint4* rp = row0 + (long long)dy * Wm * b.C8;
      for (int dx0 = 0; dx0 < f; dx0 += 8) {
        uint4 raw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = rp[(long long)(dx0 + j < f ? dx0 + j : 0) * b.C8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (dx0 + j >= f) break;
          const bf8 v = unpack8(raw[j]);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (v.v[q] > best[q]) { best[q] = v.v[q]; arg[q] = dy * f + dx0 + j; }
        }
      }
    }
    uint2 o;
    o.x = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
    o.y = (unsigned)arg[4] | ((unsigned)arg[5] << 8) | ((unsigned)arg[6] << 16) | ((unsigned)arg[7] << 24);
    b.arg[l][(((long long)n * Hl) + wy) * Wl * b.C8 + (long long)wx * b.C8 + c] = o;
  }
}

// d_avg = sum over levels of R_l^T(dout_l): finer levels sum their children, coarser levels
// route to the first maximum of avg inside the pooling window.
// MID_ / L_ >= 0: compile-time pyramid (the reference's five levels balanced at the third: the window loops unroll and a
// thread keeps its 16 + 4 + 1 loads in flight; with run-time trip counts they went out one at a time: 0.18 -> 0.07 ms)
template <int MID_, int L_>
__global__ void __launch_bounds__(TR_THREADS) balance_bwd_avg_kernel(BalBwd b) {
  const int mid = MID_ >= 0 ? MID_ : b.mid, L = L_ >= 0 ? L_ : b.L;
  const int Hm = b.H0 >> mid, Wm = b.W0 >> mid;
  const long long total = (long long)b.N * Hm * Wm * b.C8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const TrIdx4 d_ = tr_decode4(i, b.C8, Wm, Hm, rn_decode_mode(total, b.C8));
    const int c = d_.c, x = d_.x, y = d_.y, n = d_.n;
    bf8 acc;
#pragma unroll
    for (int q = 0; q < 8; ++q) acc.v[q] = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int Hl = b.H0 >> l, Wl = b.W0 >> l;
      if (l <= mid) {
        const int f = 1 << (mid - l);
#pragma unroll
        for (int dy = 0; dy < f; ++dy)
#pragma unroll
          for (int dx = 0; dx < f; ++dx) {
            const bf8 v = unpack8(b.dout[l][(((long long)n * Hl) + y * f + dy) * Wl * b.C8 +
                                            (long long)(x * f + dx) * b.C8 + c]);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc.v[q] += v.v[q];
          }
      } else {
        const int f = 1 << (l - mid);
        const int wy = y / f, wx = x / f;
        const long long oc = (((long long)n * Hl) + wy) * Wl * b.C8 + (long long)wx * b.C8 + c;
        const uint2 a2 = b.arg[l][oc];              // first maximum of avg in the window (balance_bwd_arg_kernel)
        const bf8 g = unpack8(b.dout[l][oc]);
        const unsigned me = (unsigned)((y - wy * f) * f + (x - wx * f));
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if ((((q < 4 ? a2.x : a2.y) >> (8 * (q & 3))) & 0xffu) == me) acc.v[q] += g.v[q];
      }
    }
    b.davg[i] = pack8(acc);
  }
}

// din_l = dout_l + S_l^T(d_avg / L)
template <int MID_, int L_>
__global__ void __launch_bounds__(TR_THREADS) balance_bwd_in_kernel(BalBwd b) {
  const int mid = MID_ >= 0 ? MID_ : b.mid, L = L_ >= 0 ? L_ : b.L;
  const int Hm = b.H0 >> mid, Wm = b.W0 >> mid;
  const long long total = b.begin[L];
  const float invL = 1.0f / (float)L;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = 0;
    while (i >= b.begin[l + 1]) ++l;
    const int Wl = b.W0 >> l, Hl = b.H0 >> l;
    const TrIdx4 d_ = tr_decode4(i - b.begin[l], b.C8, Wl, Hl, rn_decode_mode(total, b.C8));
    const int c = d_.c, x = d_.x, y = d_.y, n = d_.n;
    const long long o = (((long long)n * Hl) + y) * Wl * b.C8 + (long long)x * b.C8 + c;
    bf8 g = unpack8(b.dout[l][o]);
    if (l >= mid) {
      const int f = 1 << (l - mid);
      for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
          const bf8 v = unpack8(b.davg[(((long long)n * Hm) + y * f + dy) * Wm * b.C8 +
                                       (long long)(x * f + dx) * b.C8 + c]);
#pragma unroll
          for (int q = 0; q < 8; ++q) g.v[q] += v.v[q] * invL;
        }
    } else {
      const int f = 1 << (mid - l);
      const int wy = y / f, wx = x / f;
      float best[8];
      int arg[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { best[q] = -INFINITY; arg[q] = 0; }
      for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
          const bf8 v = unpack8(b.in[l][(((long long)n * Hl) + wy * f + dy) * Wl * b.C8 +
                                        (long long)(wx * f + dx) * b.C8 + c]);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (v.v[q] > best[q]) { best[q] = v.v[q]; arg[q] = dy * f + dx; }
        }
      const bf8 a = unpack8(b.davg[(((long long)n * Hm) + wy) * Wm * b.C8 + (long long)wx * b.C8 + c]);
      const int me = (y - wy * f) * f + (x - wx * f);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (arg[q] == me) g.v[q] += a.v[q] * invL;
    }
    b.din[l][o] = pack8(g);
  }
}

// scratch: d_avg bf16 [N, H0 >> mid, W0 >> mid, C], then one byte per (pixel, channel) of every level coarser than mid
extern "C" size_t rn_balance_features_bwd_scratch_bytes(int num_levels, int mid, int N, int H0, int W0, int C) {
  if (num_levels < 2 || num_levels > RN_PYR_MAX || mid < 0 || mid >= num_levels || N <= 0 || C <= 0) return 0;
  size_t b = rn_align_up((size_t)N * (H0 >> mid) * (W0 >> mid) * C * 2, 256);
  for (int l = mid + 1; l < num_levels; ++l) b += rn_align_up((size_t)N * (H0 >> l) * (W0 >> l) * C, 256);
  return b;
}

extern "C" int rn_balance_features_bwd(void* const* dout, void* const* in, void* const* din, const void* avg,
                                       void* davg_scratch, size_t scratch_bytes, int num_levels, int mid, int N, int H0,
                                       int W0, int C, void* stream) {
  RN_CHECK_ARG(dout && in && din && avg && davg_scratch && num_levels >= 2 && num_levels <= RN_PYR_MAX &&
                   mid >= 0 && mid < num_levels && C % 8 == 0,
               "rn_balance_features_bwd: bad argument");
  RN_CHECK_ARG((H0 >> mid) <= 0xffff && (1 << (num_levels - 1 - mid)) <= 16,
               "rn_balance_features_bwd: pooling windows up to 16 x 16 (one byte per window position)");
  if (scratch_bytes < rn_balance_features_bwd_scratch_bytes(num_levels, mid, N, H0, W0, C)) {
    rn_set_error("rn_balance_features_bwd: scratch too small (rn_balance_features_bwd_scratch_bytes)");
    return RN_ENOMEM;
  }
  RN_CHECK_ARG(H0 % (1 << (num_levels - 1)) == 0 && W0 % (1 << (num_levels - 1)) == 0,
               "rn_balance_features_bwd: levels must halve exactly");
  BalBwd b;
  b.L = num_levels; b.mid = mid; b.N = N; b.H0 = H0; b.W0 = W0; b.C8 = C / 8;
  b.avg = (const uint4*)avg;
  b.davg = (uint4*)davg_scratch;
  b.begin[0] = 0;
  for (int l = 0; l < num_levels; ++l) {
    RN_CHECK_ARG(dout[l] && din[l] && (l >= mid || in[l]), "rn_balance_features_bwd: null level %d", l);
    b.dout[l] = (const uint4*)dout[l];
    b.in[l] = (const uint4*)in[l];
    b.din[l] = (uint4*)din[l];
    b.begin[l + 1] = b.begin[l] + (long long)N * (H0 >> l) * (W0 >> l) * (C / 8);
  }
  hipStream_t st = (hipStream_t)stream;
  {
    char* ap = (char*)davg_scratch + rn_align_up((size_t)N * (H0 >> mid) * (W0 >> mid) * C * 2, 256);
    long long coarse = 0;
    for (int l = 0; l < RN_PYR_MAX; ++l) b.arg[l] = nullptr;
    for (int l = mid + 1; l < num_levels; ++l) {
      b.arg[l] = (uint2*)ap;
      ap += rn_align_up((size_t)N * (H0 >> l) * (W0 >> l) * C, 256);
      coarse += (long long)N * (H0 >> l) * (W0 >> l) * (C / 8);
    }
    if (coarse > 0) {
      RN_CHECK_ARG(coarse < (1ll << 31), "rn_balance_features_bwd: tensor too large");
      hipLaunchKernelGGL(balance_bwd_arg_kernel, dim3(tr_blocks(coarse)), dim3(TR_THREADS), 0, st, b);
      RN_CHECK_LAUNCH();
    }
  }
  const dim3 g_avg(tr_blocks((long long)N * (H0 >> mid) * (W0 >> mid) * (C / 8))), g_in(tr_blocks(b.begin[num_levels]));
  // the reference's only configuration: P3..P7 balanced at min_level + 1 = P4 (model/builder.py:86-89) — mid = 1.  (Round 4
  // specialised <2, 5>, "balanced at the third level": never the shape the engine runs; the generic kernels ran.)
  if (num_levels == 5 && mid == 1) {
    hipLaunchKernelGGL((balance_bwd_avg_kernel<1, 5>), g_avg, dim3(TR_THREADS), 0, st, b);
    RN_CHECK_LAUNCH();
    hipLaunchKernelGGL((balance_bwd_in_kernel<1, 5>), g_in, dim3(TR_THREADS), 0, st, b);
  } else {
    hipLaunchKernelGGL((balance_bwd_avg_kernel<-1, -1>), g_avg, dim3(TR_THREADS), 0, st, b);
    RN_CHECK_LAUNCH();
    hipLaunchKernelGGL((balance_bwd_in_kernel<-1, -1>), g_in, dim3(TR_THREADS), 0, st, b);
  }
  RN_CHECK_LAUNCH();
  return RN_OK;
}
