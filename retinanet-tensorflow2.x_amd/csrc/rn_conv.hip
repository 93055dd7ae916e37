// rn_conv.hip — K1/K2: grouped implicit-GEMM NHWC convolution on gfx950 MFMA.
// Replaces tf.keras.layers.Conv2D as used by the reference's ResNet
// (retinanet/model/backbone/resnet.py:118-144), FPN (model/neck/fpn_base.py:44-50,
// model/neck/fpn.py:47-66) and detection heads (model/head/detection_head.py:56-88).
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = R*S*Cin; A[m][k] is gathered on the fly
// from the NHWC input (for one filter tap the Cin slice of a pixel is contiguous), B is the
// prepacked weight [Cout_pad][R*S*Cin] (k contiguous), accumulation fp32 on
// v_mfma_f32_32x32x16_bf16.
//
// This file: host entry points, weight / image packing, and the 128-row kernel — workgroup = 256 threads =
// 4 wavefronts (2x2), tile BM x BN x BK = 128 x {128,64} x {64,32} — that serves the narrow layers and the
// small launches; layers with Cout >= 256 and enough tiles go to the persistent 256 x 256 x 32 kernel in
// rn_conv_big.hip (conv_use_big below).
//   * staging by `buffer_load ... lds` DMA (16 B per lane, no VGPR round trip), two LDS stages, ONE
//     barrier per K step: the DMA of step t+1 is issued before the MFMAs of step t.
//   * LDS tiles are [rows][BK] bf16 with the 16-byte slot index XOR-swizzled by
//     (row / rows_per_256B) — applied to the DMA's per-lane source chunk and to the ds_read_b128
//     fragment address — so the fragment reads (one row per lane, same k slot) hit 16 distinct
//     16-byte slots of the 256-byte bank row: conflict free.
//   * padding / image borders / M tail: per-row tap-validity bitmask computed once per tile;
//     invalid pieces get an out-of-range buffer offset, which the hardware zero-fills.
//   * epilogue: acc*scale[c]+shift[c] staged through LDS as fp32 [BM][BN] (reusing the A/B
//     buffers), then read back row-contiguous: + residual, activation, convert, 8/16-byte
//     coalesced stores.  BN(+bias) folding makes Conv+BN+ReLU(+add) one kernel at inference.
//   * grouped launch: up to 10 independent segments (pyramid levels / both heads) share one
//     grid, so the small P5-P7 problems ride along with P3 instead of under-filling 256 CUs.
//   * blockIdx -> tile map is XCD aware: consecutive tiles (same A rows, neighbouring n tiles)
//     land on the same XCD's L2 (blocks are dispatched round-robin over the 8 XCDs).
// Roofline: MFMA-bound for K >= ~512; the small-K 1x1 convs of ResNet stage 1-2 are HBM-bound.
#include <map>
#include <mutex>
#include <tuple>

#include <algorithm>
#include "rn_conv_dev.h"

// ABL: ablation mask for tools/bench_conv.py (0 in production): 1 = B tile loaded once,
// 2 = A tile loaded once, 4 = no MFMA.
// WM x WN wavefronts (64 x (BN/WN) wave tiles), two LDS stages: 128 x {128,64} x {64,32}, 2x2 waves
// (64 KB LDS, 2 workgroups/CU) for the narrow / small launches; the wide layers go to rn_conv_big.hip.
// SPLIT (rnet_hip.h: rn_conv_problem.splitk_ws; small launches of deep layers — batch-1 / batch-8 inference): every tile is
// cut along K into args.split_s parts, one workgroup each (grid = total_tiles * split_s, the parts of a tile neighbours in
// the XCD-aware numbering).  A part writes its raw fp32 accumulators — in the [BM][BN] row layout the epilogue's second
// stage reads — to its slot of the workspace with write-through (sc1) stores, drains them, and counts itself in on the
// tile's counter with a returning agent-scope atomic; the part that arrives LAST (whichever it is: nobody waits) reads
// all slots back with sc1 loads, adds them IN PART ORDER (the same bits on every run), zeroes the counter and runs the
// epilogue arithmetic (bias, rounding points, BatchNorm affine, residual, activation) on the sums.
template <int BM, int BN, int BK, bool OUT_F32, int ABL = 0, int WM = 2, int WN = 2, int STAGES = 2, bool SPLIT = false>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_fwd_kernel(const ConvArgs args) {
  constexpr int NWAVES = WM * WN;
  constexpr int NTHREADS = 64 * NWAVES;
  constexpr int SLOTS = BK / 8;
  constexpr int RPI = 64 / SLOTS;                   // rows per wave DMA instruction (1 KiB)
  constexpr int A_INSTR = BM / RPI / NWAVES;        // DMA instructions per wave for the A tile
  constexpr int B_INSTR = BN / RPI / NWAVES;
  constexpr int WTM = BM / WM, WTN = BN / WN;       // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int KSUB = BK / 16;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static_assert(A_INSTR >= 1 && B_INSTR >= 1, "tile too small for the wave count");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- tile lookup (XCD-aware remap of the linear block id) ---------------------------------
  int tile, part = 0;
  {
    const int total = SPLIT ? args.vtotal : args.total_tiles;   // work units: tiles, or (tile, part) pairs
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    if (SPLIT) {
      const int v = tile;
      tile = v / args.split_s;
      part = v - tile * args.split_s;
    }
  }
  int si = 0;
#pragma unroll 1
  for (int i = 1; i < args.nseg; ++i)
    if (tile >= args.seg[i].tile_begin) si = i;
  const ConvSegDev& sg = args.seg[si];
  const int lt = tile - sg.tile_begin;
  const int m_tile = lt / sg.n_tiles, n_tile = lt - m_tile * sg.n_tiles;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN, wave_n = wave % WN;

  const int R = args.R, S = args.S;
  const int H = sg.H, W = sg.W, Cin = sg.CinP, PS = sg.pix_stride;
  const int M = sg.M;
  const int Ktot = R * S * Cin;
  const int cwrap = sg.cwrap;   // split-bf16 weight planes: the input channels repeat every cwrap

  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.x, 0, (int)((long long)sg.N * H * W * PS * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.w, 0, (int)((long long)sg.n_tiles * BN * Ktot * 2), 0x00020000);

  // ---- per-lane DMA bookkeeping: instruction j of this wave fills rows (j*4 + wave)*RPI .. ----
  const int d_row = lane / SLOTS, d_pos = lane % SLOTS;
  unsigned a_off[A_INSTR];   // byte offset of (row's first tap pixel, its k chunk) or RN_OOB
  unsigned a_mask[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (j * NWAVES + wave) * RPI + d_row;
    const int chunk = d_pos ^ lds_swz<BK>(row);
    const int m = m0 + row;
    const int mm = m < M ? m : 0;
    int ox, oy, n;
    if (args.pad_ & 1) {   // every M < 2^22: float-reciprocal divisions (~8 VALU instead of ~45 each; four rows x two per thread
                           // were ~1 us of every launch's prologue — a tenth of a batch-1 layer)
      const int t2 = rn_fdiv(mm, sg.Wo, rn_rcp((float)sg.Wo));
      ox = mm - t2 * sg.Wo;
      n = rn_fdiv(t2, sg.Ho, rn_rcp((float)sg.Ho));
      oy = t2 - n * sg.Ho;
    } else {
      ox = mm % sg.Wo;
      const int t2 = mm / sg.Wo;
      oy = t2 % sg.Ho;
      n = t2 / sg.Ho;
    }
    const int iy0 = oy * args.sh - args.pt, ix0 = ox * args.sw - args.pl;
    a_off[j] = (unsigned)(((((long long)n * H + iy0) * W + ix0) * PS + chunk * 8) * 2);
    unsigned mask = 0;
    if (m < M) {
      for (int r = 0; r < R; ++r)
        for (int s = 0; s < S; ++s) {
          const bool ok = (unsigned)(iy0 + r) < (unsigned)H && (unsigned)(ix0 + s) < (unsigned)W;
          mask |= (ok ? 1u : 0u) << (r * S + s);
        }
    }
    a_mask[j] = mask;
  }
  unsigned b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int row = (j * NWAVES + wave) * RPI + d_row;
    const int chunk = d_pos ^ lds_swz<BK>(row);
    b_off[j] = (unsigned)(((long long)(n0 + row) * Ktot + chunk * 8) * 2);
  }

  // fragment read offsets (row = lane&31 of a 32-row tile, k slot = 2*kk + (lane>>5))
  const int frag_row = lane & 31, frag_half = lane >> 5;
  int rd_a[TM][KSUB], rd_b[TN][KSUB];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk)
      rd_a[i][kk] = lds_slot_off<BK>(wave_m * WTM + i * 32 + frag_row, kk * 2 + frag_half);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk)
      rd_b[j][kk] = A_BYTES + lds_slot_off<BK>(wave_n * WTN + j * 32 + frag_row, kk * 2 + frag_half);

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps_all = R * S * (Cin / BK);
  // K steps [kbeg, ksteps) of this workgroup: all of them, or the part's share
  const int kbeg = SPLIT ? (int)((long long)part * ksteps_all / args.split_s) : 0;
  const int ksteps = SPLIT ? (int)((long long)(part + 1) * ksteps_all / args.split_s) : ksteps_all;

#define RN_ISSUE_TILE(buf, tap_, c0_)                                                         \
  do {                                                                                        \
    const int r__ = (tap_) / S, s__ = (tap_) - r__ * S;                                       \
    const int cw__ = (c0_) < cwrap ? (c0_) : ((c0_) < 2 * cwrap ? (c0_) - cwrap : (c0_) - 2 * cwrap);   \
    const unsigned tap_off__ = (unsigned)((((long long)r__ * W + s__) * PS + cw__) * 2);      \
    char* st__ = smem + (buf) * STAGE_BYTES;                                                  \
    if (!(ABL & 2) || ((tap_) == 0 && (c0_) == 0)) {                                          \
      _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) {                                   \
        const unsigned v__ = ((a_mask[j] >> (tap_)) & 1u) ? a_off[j] + tap_off__ : RN_OOB;    \
        dma16(rs_x, st__ + (j * NWAVES + wave) * 1024, v__);                                       \
      }                                                                                       \
    }                                                                                         \
    const unsigned koff__ = (unsigned)(((long long)(tap_) * Cin + (c0_)) * 2);                \
    if (!(ABL & 1) || ((tap_) == 0 && (c0_) == 0)) {                                          \
      _Pragma("unroll") for (int j = 0; j < B_INSTR; ++j)                                     \
        dma16(rs_w, st__ + A_BYTES + (j * NWAVES + wave) * 1024, b_off[j] + koff__);               \
    }                                                                                         \
  } while (0)

  int tap = 0, c0 = 0;   // coordinates of the NEXT tile to issue
  if (SPLIT) {
    const int cpk = Cin / BK;
    tap = kbeg / cpk;
    c0 = (kbeg - tap * cpk) * BK;
  }
#define RN_ADVANCE()  \
  do {                \
    c0 += BK;         \
    if (c0 >= Cin) {  \
      c0 = 0;         \
      ++tap;          \
    }                 \
  } while (0)

  if (ABL & 32) {
    // ablation: no main loop at all (launch + prologue + epilogue cost)
  } else if (STAGES == 2) {
    RN_ISSUE_TILE(0, tap, c0);
    RN_ADVANCE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int kt = kbeg; kt < ksteps; ++kt) {
      if (kt + 1 < ksteps) {
        RN_ISSUE_TILE(cur ^ 1, tap, c0);
        RN_ADVANCE();
      }
      const char* base = smem + cur * STAGE_BYTES;
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        bf16x8_t fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8_t*)(base + rd_a[i][kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8_t*)(base + rd_b[j][kk]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            if (!(ABL & 4)) {
              acc[i][j] = RN_MFMA_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            } else {
              acc[i][j][0] += (float)fa[i][0] + (float)fb[j][0];
            }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  }
  else if (STAGES >= 3 && STAGES <= 8) {
    // STAGES LDS stages, counted waits: the DMA of step kt + STAGES - 1 is issued while step kt is multiplied, so a tile's
    // bytes have STAGES - 1 steps to arrive instead of one (the two-stage loop waits for ALL outstanding pieces — i.e. for
    // the round trip of the tile it issued at the top of the same step — before every barrier: ~1.2 us per K step whatever
    // the step computes).  One s_barrier per step (no fence: __syncthreads() would put s_waitcnt vmcnt(0) in front of it):
    // behind it every wave's pieces of step kt have landed and every wave has finished reading step kt - 1, whose stage
    // the next issue overwrites.
    constexpr int PIECES = A_INSTR + B_INSTR;   // DMA instructions per wave and K step
    constexpr int AHEAD = STAGES - 1;
#pragma unroll
    for (int q = 0; q < AHEAD; ++q)
      if (kbeg + q < ksteps) {
        RN_ISSUE_TILE(q, tap, c0);
        RN_ADVANCE();
      }
    int cur = 0;
#pragma unroll 1
    for (int kt = kbeg; kt < ksteps; ++kt) {
      // everything but the pieces of the AHEAD - 1 newest steps has landed (at the tail, where fewer are in flight: all)
      if (kt + AHEAD - 1 < ksteps) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1) * PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + AHEAD < ksteps) {
        const int nxt = cur >= 1 ? cur - 1 : STAGES - 1;   // (cur + STAGES - 1) % STAGES
        RN_ISSUE_TILE(nxt, tap, c0);
        RN_ADVANCE();
      }
      const char* base = smem + cur * STAGE_BYTES;
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        bf16x8_t fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8_t*)(base + rd_a[i][kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8_t*)(base + rd_b[j][kk]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = RN_MFMA_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      cur = cur == STAGES - 1 ? 0 : cur + 1;
    }
    __syncthreads();   // the epilogue reuses the stages
  }
  else if (STAGES == 32) {
    // 128 x 128 x 64 (32 KB per K step; three whole stages would leave one workgroup per CU): THREE stages of pixels, TWO of
    // weights, 80 KB — two workgroups per CU.  The pixels come from HBM (or another XCD's L2) and get two steps to arrive; the
    // weights are L2-resident and get one.  Issue order per step: weights of step kt + 1, THEN pixels of step kt + 2, so that
    // "everything but the last A_INSTR pieces" (loads retire in order) = pixels and weights of step kt + 1.
    constexpr int B0 = 3 * A_BYTES;   // the weight stages sit behind the three pixel stages
    int tap_a = tap, c0_a = c0, tap_b = tap, c0_b = c0;   // the pixel stream runs one step ahead of the weight stream
#define RN_ISSUE_A(buf, tap_, c0_)                                                            \
  do {                                                                                        \
    const int r__ = (tap_) / S, s__ = (tap_) - r__ * S;                                       \
    const int cw__ = (c0_) < cwrap ? (c0_) : ((c0_) < 2 * cwrap ? (c0_) - cwrap : (c0_) - 2 * cwrap);   \
    const unsigned tap_off__ = (unsigned)((((long long)r__ * W + s__) * PS + cw__) * 2);      \
    _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) {                                     \
      const unsigned v__ = ((a_mask[j] >> (tap_)) & 1u) ? a_off[j] + tap_off__ : RN_OOB;      \
      dma16(rs_x, smem + (buf) * A_BYTES + (j * NWAVES + wave) * 1024, v__);                   \
    }                                                                                         \
  } while (0)
#define RN_ISSUE_B(buf, tap_, c0_)                                                            \
  do {                                                                                        \
    const unsigned koff__ = (unsigned)(((long long)(tap_) * Cin + (c0_)) * 2);                \
    _Pragma("unroll") for (int j = 0; j < B_INSTR; ++j)                                       \
      dma16(rs_w, smem + B0 + (buf) * B_BYTES + (j * NWAVES + wave) * 1024, b_off[j] + koff__); \
  } while (0)
#define RN_ADV(tap_, c0_) do { c0_ += BK; if (c0_ >= Cin) { c0_ = 0; ++tap_; } } while (0)
    // prologue: pixels of steps 0 and 1, weights of step 0 (weights first where both are issued)
    RN_ISSUE_B(0, tap_b, c0_b); RN_ADV(tap_b, c0_b);
    RN_ISSUE_A(0, tap_a, c0_a); RN_ADV(tap_a, c0_a);
    if (kbeg + 1 < ksteps) { RN_ISSUE_A(1, tap_a, c0_a); RN_ADV(tap_a, c0_a); }
    int cur_a = 0, cur_b = 0;
#pragma unroll 1
    for (int kt = kbeg; kt < ksteps; ++kt) {
      // pixels and weights of step kt have landed: everything but the pixel pieces of step kt + 1 (the newest issue)
      if (kt + 1 < ksteps) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(A_INSTR) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 1 < ksteps) { RN_ISSUE_B(cur_b ^ 1, tap_b, c0_b); RN_ADV(tap_b, c0_b); }
      if (kt + 2 < ksteps) {
        const int nxt = cur_a >= 1 ? cur_a - 1 : 2;   // (cur_a + 2) % 3
        RN_ISSUE_A(nxt, tap_a, c0_a); RN_ADV(tap_a, c0_a);
      }
      const char* base_a = smem + cur_a * A_BYTES;
      const char* base_b = smem + B0 + cur_b * B_BYTES - A_BYTES;   // rd_b carries + A_BYTES
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        bf16x8_t fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const bf16x8_t*)(base_a + rd_a[i][kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *(const bf16x8_t*)(base_b + rd_b[j][kk]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = RN_MFMA_32x32x16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      cur_a = cur_a == 2 ? 0 : cur_a + 1;
      cur_b ^= 1;
    }
    __syncthreads();   // the epilogue reuses the stages
#undef RN_ADV
#undef RN_ISSUE_B
#undef RN_ISSUE_A
  }
#undef RN_ADVANCE
#undef RN_ISSUE_TILE

  // ---- epilogue ------------------------------------------------------------------------------
  if (ABL & 16) {   // ablation: no epilogue (one never-taken store keeps the accumulators alive)
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) ((float*)sg.y)[tid] = t;
    return;
  }
  // fused BatchNorm backward reduction (rn_conv_segment.bn_bwd_y): the y values stage 2 needs — this thread's 4
  // channels of its BM / ROWS rows — are fetched NOW, so that their HBM latency runs under stage 1 and the barrier
  // (loaded where they are used, behind the stores of the previous row, each launch paid ~35 us for them)
  constexpr int E_TPR = BN / 4, E_ROWS = NTHREADS / E_TPR, E_NIT = BM / E_ROWS;
  uint2 ypre[E_NIT];
  if (!OUT_F32 && sg.bn_partial != nullptr && sg.bn_y != nullptr) {
    const int er_ = tid / E_TPR, n_ = n0 + (tid % E_TPR) * 4;
#pragma unroll
    for (int it = 0; it < E_NIT; ++it) {
      const int m_ = m0 + er_ + it * E_ROWS;
      ypre[it] = make_uint2(0u, 0u);
      if (m_ < M && n_ < sg.Cout) ypre[it] = *(const uint2*)(sg.bn_y + (long long)m_ * sg.Cout + n_);
    }
  }
  // stage 1: registers -> LDS fp32 [BM][BN]: conv (+bias) output and BatchNorm affine, each rounded to bf16 where
  // the reference holds a bf16 tensor between two layers (rnet_hip.h, rn_conv_segment)
  float* cl = (float*)smem;
  const int Cout = sg.Cout;
  const bool affine = sg.scale != nullptr || sg.shift != nullptr;
  const bool has_res = sg.residual != nullptr;
  const bool round1 = !OUT_F32 && (affine || has_res);
  const bool round2 = !OUT_F32 && affine && has_res;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int nl = wave_n * WTN + j * 32 + (lane & 31);
    const int n = n0 + nl;
    float sc = 1.0f, sf = 0.0f, bs = 0.0f;
    if (n < Cout) {
      if (sg.scale) sc = sg.scale[n];
      if (sg.shift) sf = sg.shift[n];
      if (sg.bias) bs = sg.bias[n];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float t[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) t[r] = SPLIT ? acc[i][j][r] : acc[i][j][r] + bs;   // SPLIT: raw partial sums
      if (!SPLIT && round1) {   // uniform branches per stage, not per element
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = rn_rb(t[r]);
      }
      if (!SPLIT && affine) {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = t[r] * sc + sf;
      }
      if (!SPLIT && round2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = rn_rb(t[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wave_m * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        cl[ml * BN + nl] = t[r];
      }
    }
  }
  __syncthreads();
  // stage 2: row-contiguous read back, residual, activation, store
  constexpr int TPR = BN / 4;                 // threads per row (4 channels each)
  constexpr int ROWS = NTHREADS / TPR;        // rows per pass
  const int er = tid / TPR, ec = (tid % TPR) * 4;
  const int n = n0 + ec;
  // ---- SPLIT: partial tile -> workspace, count in; only the last part to arrive carries on (see the kernel's header) ----
  typedef unsigned cf_u32x4_t __attribute__((ext_vector_type(4)));
  constexpr int SLOT_BYTES = BM * BN * 4;
  __amdgpu_buffer_rsrc_t rs_parts = __builtin_amdgcn_make_buffer_rsrc((void*)sg.y, 0, 0, 0x00020000);   // (set below; unused unless SPLIT)
  float ssc[4] = {1.f, 1.f, 1.f, 1.f}, ssf[4] = {0.f, 0.f, 0.f, 0.f}, sbs[4] = {0.f, 0.f, 0.f, 0.f};
  if (SPLIT) {
    const int S_ = args.split_s;
    char* const slot0 = (char*)args.ws + RN_SPLITK_HEADER_BYTES + (size_t)tile * S_ * SLOT_BYTES;
    const __amdgpu_buffer_rsrc_t rs_mine =
        __builtin_amdgcn_make_buffer_rsrc((void*)(slot0 + (size_t)part * SLOT_BYTES), 0, SLOT_BYTES, 0x00020000);
#pragma unroll
    for (int it = 0; it < E_NIT; ++it) {
      const int rr = er + it * ROWS;
      const float4 v = *(const float4*)(cl + rr * BN + ec);
      const cf_u32x4_t u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      __builtin_amdgcn_raw_buffer_store_b128(u, rs_mine, (rr * BN + ec) * 4, 0, 16);   // aux 16 = sc1: write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // every thread's stores have drained; nobody reads cl any more
    if (tid == 0) {
      unsigned* const cnt = (unsigned*)args.ws + tile;
      const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)(S_ - 1)) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
      *(volatile unsigned*)smem = old;
    }
    __syncthreads();
    if (*(volatile unsigned*)smem != (unsigned)(S_ - 1)) return;
    __syncthreads();            // (the statistics reduction below reuses smem)
    rs_parts = __builtin_amdgcn_make_buffer_rsrc((void*)slot0, 0, S_ * SLOT_BYTES, 0x00020000);
    if (n < sg.Cout) {          // this thread's four channels: the arithmetic stage 1 skipped
      if (sg.scale) { const float4 a = *(const float4*)(sg.scale + n); ssc[0] = a.x; ssc[1] = a.y; ssc[2] = a.z; ssc[3] = a.w; }
      if (sg.shift) { const float4 a = *(const float4*)(sg.shift + n); ssf[0] = a.x; ssf[1] = a.y; ssf[2] = a.z; ssf[3] = a.w; }
      if (sg.bias) { const float4 a = *(const float4*)(sg.bias + n); sbs[0] = a.x; sbs[1] = a.y; sbs[2] = a.z; sbs[3] = a.w; }
    }
  }
  // fused BatchNorm forward statistics (rn_conv_segment.bn_partial): sums of the STORED bf16 values of this
  // thread's 4 channels over its rows; reduced over the workgroup's 128 rows below
  const bool stats = !OUT_F32 && sg.bn_partial != nullptr;
  // ... or stage 1 of the BatchNorm BACKWARD reduction of the layer whose dz this launch writes
  // (rn_conv_segment.bn_bwd_y): sums of g = dz * [y*scale + shift > 0] and of g*xhat
  const bool bnbwd = stats && sg.bn_y != nullptr;
  float bsc[4] = {0.f, 0.f, 0.f, 0.f}, bsh[4] = {0.f, 0.f, 0.f, 0.f}, bmu[4] = {0.f, 0.f, 0.f, 0.f}, bis[4] = {0.f, 0.f, 0.f, 0.f};
  if (bnbwd && n < Cout) {
    const float4 m4 = *(const float4*)(sg.bn_fwd + n), i4 = *(const float4*)(sg.bn_fwd + Cout + n);
    const float4 a4 = *(const float4*)(sg.bn_fwd + 2 * Cout + n), b4 = *(const float4*)(sg.bn_fwd + 3 * Cout + n);
    bmu[0] = m4.x; bmu[1] = m4.y; bmu[2] = m4.z; bmu[3] = m4.w;
    bis[0] = i4.x; bis[1] = i4.y; bis[2] = i4.z; bis[3] = i4.w;
    bsc[0] = a4.x; bsc[1] = a4.y; bsc[2] = a4.z; bsc[3] = a4.w;
    bsh[0] = b4.x; bsh[1] = b4.y; bsh[2] = b4.z; bsh[3] = b4.w;
  }
  float st0[4] = {0.f, 0.f, 0.f, 0.f}, st1[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < Cout) {
#pragma unroll
    for (int it = 0; it < E_NIT; ++it) {
      const int rr = er + it * ROWS;
      const int m = m0 + rr;
      if (m >= M) break;
      float4 v;
      if (SPLIT) {
        // all parts' loads in flight before the first add; parts past split_s lie outside the descriptor and read as
        // zeros (at most 8 parts: the host's plan)
        cf_u32x4_t pv[8];
#pragma unroll
        for (int p_ = 0; p_ < 8; ++p_)
          pv[p_] = __builtin_amdgcn_raw_buffer_load_b128(rs_parts, (rr * BN + ec) * 4 + p_ * SLOT_BYTES, 0, 16);   // sc1
        float f4[4] = {__uint_as_float(pv[0].x), __uint_as_float(pv[0].y), __uint_as_float(pv[0].z), __uint_as_float(pv[0].w)};
#pragma unroll
        for (int p_ = 1; p_ < 8; ++p_) {
          f4[0] += __uint_as_float(pv[p_].x); f4[1] += __uint_as_float(pv[p_].y);
          f4[2] += __uint_as_float(pv[p_].z); f4[3] += __uint_as_float(pv[p_].w);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float t = f4[q] + sbs[q];
          if (round1) t = rn_rb(t);
          if (affine) t = t * ssc[q] + ssf[q];
          if (round2) t = rn_rb(t);
          f4[q] = t;
        }
        v = make_float4(f4[0], f4[1], f4[2], f4[3]);
      } else {
        v = *(const float4*)(cl + rr * BN + ec);
      }
      const long long o = (long long)m * Cout + n;
      if (sg.residual) {
        const uint2 rv = *(const uint2*)(sg.residual + o);
        v.x += rn_bf16_to_f32((uint16_t)(rv.x & 0xffffu));
        v.y += rn_bf16_to_f32((uint16_t)(rv.x >> 16));
        v.z += rn_bf16_to_f32((uint16_t)(rv.y & 0xffffu));
        v.w += rn_bf16_to_f32((uint16_t)(rv.y >> 16));
      }
      {
        float f4[4] = {v.x, v.y, v.z, v.w};
        if (!OUT_F32 && args.act == RN_ACT_SWISH) {   // tf.nn.swish reads the layer output as a bf16 tensor
#pragma unroll
          for (int q = 0; q < 4; ++q) f4[q] = rn_rb(f4[q]);
        }
        rn_apply_act_n<4>(f4, args.act);
        v = make_float4(f4[0], f4[1], f4[2], f4[3]);
      }
      if ((ABL & 8) && v.x != 123.456f) continue;   // ablation: no global stores
      if (OUT_F32) {
        *(float4*)((float*)sg.y + o) = v;
      } else {
        uint2 pk;
        pk.x = rn_pack_bf16x2(v.x, v.y);
        pk.y = rn_pack_bf16x2(v.z, v.w);
        *(uint2*)((uint16_t*)sg.y + o) = pk;
        if (stats) {
          const float w4[4] = {rn_lo16(pk.x), rn_hi16(pk.x), rn_lo16(pk.y), rn_hi16(pk.y)};
          if (bnbwd) {
            const uint2 yv = ypre[it];
            const float y4[4] = {rn_lo16(yv.x), rn_hi16(yv.x), rn_lo16(yv.y), rn_hi16(yv.y)};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float g = (y4[q] * bsc[q] + bsh[q]) > 0.0f ? w4[q] : 0.0f;
              st0[q] += g;
              st1[q] += g * ((y4[q] - bmu[q]) * bis[q]);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) { st0[q] += w4[q]; st1[q] += w4[q] * w4[q]; }
          }
        }
      }
    }
  }
  if (stats) {   // uniform per workgroup (one segment per tile)
    __syncthreads();                       // every thread is done reading cl
    float* red = cl;                       // [2][ROWS][BN]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[er * BN + ec + q] = st0[q];
      red[(ROWS + er) * BN + ec + q] = st1[q];
    }
    __syncthreads();
    if (tid < BN && n0 + tid < Cout) {     // fixed order over the ROWS row lanes: deterministic
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) { s0 += red[r * BN + tid]; s1 += red[(ROWS + r) * BN + tid]; }
      float* dst = sg.bn_partial + (long long)(m0 / BM) * 2 * Cout + n0 + tid;
      dst[0] = s0;
      dst[Cout] = s1;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------
extern "C" int rn_conv_cout_pad(int Cout) { return Cout <= 64 ? 64 : (int)rn_align_up((size_t)Cout, 128); }
// packed-weight channel count: Cin rounded up to the K step (32 below 64 channels, 64 above)
extern "C" int rn_conv_cin_pad(int Cin) { return Cin <= 32 ? 32 : (int)rn_align_up((size_t)Cin, 64); }

template <int BM, int BN, int BK, bool F32, bool SPLIT = false>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int stage = (BM + BN) * BK * 2;
  constexpr int epi = BM * BN * 4;
  // as many stages (up to four) as leave two workgroups per CU: 128 x 64 x 64: three (72 KB); 128 x 128 x 32: four (64 KB);
  // 128 x 64 x 32: four (48 KB); 128 x 128 x 64 keeps two (three would be 96 KB, one workgroup per CU: measured 35 - 45 %
  // slower wherever a launch has more tiles than compute units, equal below)
  // (32 = three stages of pixels + two of weights, 80 KB, for the 32 KB steps of 128 x 128 x 64)
  constexpr int STAGES = (4 * stage <= 72 * 1024) ? 4 : ((3 * stage <= 72 * 1024) ? 3 : ((BM == 128 && BN == 128 && BK == 64) ? 32 : 2));
  static const bool two_stages = getenv("RNET_CONV128_STAGES") && atoi(getenv("RNET_CONV128_STAGES")) == 2;   // A/B probe
  if (STAGES >= 3 && two_stages) {
    constexpr int lds2 = (2 * stage > epi) ? 2 * stage : epi;
    auto kern2 = conv_fwd_kernel<BM, BN, BK, F32, 0, 2, 2, 2, SPLIT>;
    if (lds2 > 48 * 1024)
      RN_CHECK_HIP(hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
    hipLaunchKernelGGL(kern2, dim3(SPLIT ? a.vtotal : a.total_tiles), dim3(256), lds2, st, a);
    RN_CHECK_LAUNCH();
    return RN_OK;
  }
  constexpr int ring = STAGES == 32 ? 3 * BM * BK * 2 + 2 * BN * BK * 2 : STAGES * stage;
  constexpr int lds = (ring > epi) ? ring : epi;
  auto kern = conv_fwd_kernel<BM, BN, BK, F32, 0, 2, 2, STAGES, SPLIT>;
  if (lds > 48 * 1024)
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(SPLIT ? a.vtotal : a.total_tiles), dim3(256), lds, st, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
// the 128-row kernel by tile shape / output type / split
template <bool SPLIT>
static int launch_conv128(const ConvArgs& a, int BN, int BK, bool f32, hipStream_t st) {
  if (BN == 128 && BK == 64) return f32 ? launch_conv<128, 128, 64, true, SPLIT>(a, st) : launch_conv<128, 128, 64, false, SPLIT>(a, st);
  if (BN == 64 && BK == 64) return f32 ? launch_conv<128, 64, 64, true, SPLIT>(a, st) : launch_conv<128, 64, 64, false, SPLIT>(a, st);
  if (BN == 128 && BK == 32) return f32 ? launch_conv<128, 128, 32, true, SPLIT>(a, st) : launch_conv<128, 128, 32, false, SPLIT>(a, st);
  return f32 ? launch_conv<128, 64, 32, true, SPLIT>(a, st) : launch_conv<128, 64, 32, false, SPLIT>(a, st);
}

template <int ABL>
static int launch_ablate(const ConvArgs& a, hipStream_t st) {
  constexpr int lds = 2 * (128 + 128) * 64 * 2;
  auto kern = conv_fwd_kernel<128, 128, 64, false, ABL>;
  RN_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(a.total_tiles), dim3(CONV_THREADS), lds, st, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// 256 x 256 x 32 tiles (rn_conv_big.hip) for the MFMA-bound layers: every segment at least 256 output
// channels wide, and enough tiles to fill the 256 CUs (one workgroup per CU) a few times over.
// (rn_launch_opts: conv_tile forces either family, conv_big_min_tiles moves the threshold.)
// Last-round split of a persistent launch (rnet_hip.h: rn_conv_problem.splitk_ws).  G0 workgroups walk `total` tiles in
// rounds; the L = total mod G0 tiles of the last round are each cut into S = min(G0 / L, chunks / 4, 4) parts along K, so
// that round keeps L * S workgroups busy for 1/S of a tile (+ the exchange of L * (S - 1) accumulator tiles through the
// workspace) instead of L workgroups for a whole one.
// GEMM columns of a segment: its output channels, or for w_pair (two weight planes along Cout) the packed rows
extern "C" int rn_conv_pair_rows(int Cout) { return Cout > 0 ? 128 * ((Cout + 63) / 64) : 0; }
static int seg_cols(const rn_conv_segment& s) { return s.w_pair ? rn_conv_pair_rows(s.Cout) : s.Cout; }
// channel granularity of the 256- / 512-row epilogues: 16-byte rows of bf16, or of f32 (float4 stores)
static bool seg_cout_ok(const rn_conv_problem* p, const rn_conv_segment& s) {
  return s.Cout % (p->out_dtype == RN_DT_F32 && s.w_pair ? 4 : 8) == 0;
}

static int splitk_parts(int total, int min_chunks, int G0, long long* bytes) {
  const int L = total % G0;
  if (bytes) *bytes = 0;
  if (L == 0 || L > 511 || min_chunks < 8) return 1;   // 8 arrival counters per leftover tile in a 4096-word header
  // every part costs its tile one more 256 KB slot to write and part 0 one more to read (~4 us each): at least 4 chunks
  // (36 K steps, ~20 us) per part, at most 4 parts
  int S = G0 / L;
  if (S > min_chunks / 4) S = min_chunks / 4;
  if (S > 4) S = 4;
  if (S < 2) return 1;
  if (bytes) *bytes = RN_SPLITK_HEADER_BYTES + (long long)L * S * RN_SPLITK_SLOT_BYTES;   // one slot per part
  return S;
}

static bool conv_halo_shape(const rn_conv_problem* p, int BM);
static int conv_min_chunks(const rn_conv_problem* p);

static bool conv_use_big(const rn_conv_problem* p) {
  if (p->opts.conv_tile == 1 || p->opts.conv_tile == 3) return false;
  long long tiles256 = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_conv_segment& s = p->seg[i];
    if (rn_conv_cout_pad(seg_cols(s)) < 256 || !seg_cout_ok(p, s)) return false;
    if (s.bias && s.residual) return false;   // the residual variants of the 256-row kernels carry no bias path
    tiles256 += rn_cdiv((long long)s.N * s.Ho * s.Wo, 256) * rn_cdiv(seg_cols(s), 256);
  }
  if (p->opts.conv_tile == 2 || tiles256 >= (p->opts.conv_big_min_tiles > 0 ? p->opts.conv_big_min_tiles : 192)) return true;
  // A launch of fewer tiles than compute units (batch-8 inference, ResNet stage 3 / 4) stays on the 128-row kernel.  Round 4
  // sent it to the halo kernel's 256 x 256 tiles, every tile cut along K, when a split-K workspace was attached; with the
  // counted-wait K loop of the 128-row kernel (and its own split-K) that choice loses: stage-3 3x3 at batch 8 43.7 vs 36.0 us,
  // batch 16 53.2 vs 46.1; stage-4 3x3 45.3 vs 42.6 at batch 8, 59.4 vs 61.4 at batch 16 (tools/probes/ab_small_3x3.sh); the
  // five-level head / FPN launches at batch 1: 46.6 vs 43.9, 63.9 vs 47.0, 41.8 vs 32.9 (tools/probes/ab_b1_heads.sh).
  // rn_launch_opts.conv_tile = 2 still takes a small launch to the halo kernel, split when a workspace is attached.
  return false;
}

// 3x3 / stride 1 / pad 1 launches of the 256-row class go to the halo kernel (rn_conv_halo.hip) when every
// segment's worst tile fits its patch buffer.
static bool conv_halo_shape(const rn_conv_problem* p, int BM) {   // BM: pixels per tile (256, or 512 for the narrow form)
  if (p->opts.conv_no_halo) return false;
  if (p->R != 3 || p->S != 3 || p->stride_h != 1 || p->stride_w != 1 || p->pad_top != 1 || p->pad_left != 1)
    return false;
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int>, int> patch_px;   // (N, H, W, pitch * 1024 + BM) -> worst patch, computed once
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_conv_segment& s = p->seg[i];
    if (s.Ho != s.H || s.Wo != s.W || s.Cin % 32 != 0) return false;
    if ((long long)s.N * s.H * s.W >= (1ll << 22)) return false;   // the kernel's float-reciprocal divisions
    int px;
    {
      std::lock_guard<std::mutex> lock(mu);
      // the wide pitch when its patch fits, else the tight one (the kernel is told per segment: halo_pitch)
      for (int pitch : {rn_conv_halo_pitch(s.W), s.W + 1}) {
        auto key = std::make_tuple(s.N, s.H, s.W, pitch * 1024 + BM);
        auto it = patch_px.find(key);
        if (it == patch_px.end()) it = patch_px.emplace(key, rn_conv_halo_patch_pixels(s.N, s.H, s.W, pitch, BM)).first;
        px = it->second;
        if (px <= rn_conv_halo_capacity(BM)) break;
      }
    }
    if (px > rn_conv_halo_capacity(BM)) return false;
  }
  return true;
}

// 3x3 / stride 1 / pad 1 layers with 64 < Cout <= 128 (ResNet stage 2: 128 -> 128 at 80 x 80, forward and data gradient):
// the halo kernel with 512 x 128 tiles (rn_conv_halo.hip, HaloGeo<4>: 4 x 2 waves of 128 pixels x 64 channels).  On the
// 128-row kernel such a layer staged its pixels once per tap (the LDS-DMA path bound it at ~550 TFLOP/s); in a 256-wide
// tile of the halo kernel half the waves multiply zero rows.
static bool conv_use_halo512(const rn_conv_problem* p) {
  if (p->opts.conv_tile == 1 || p->opts.conv_no_halo) return false;
  long long tiles = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_conv_segment& s = p->seg[i];
    const int cp = rn_conv_cout_pad(seg_cols(s));
    // conv_tile = 3: any width, also <= 64 channels (half of the tile's columns are then zero weights)
    if ((cp % 128 != 0 && !(cp == 64 && p->opts.conv_tile == 3)) || (cp != 128 && p->opts.conv_tile != 3) || !seg_cout_ok(p, s))
      return false;
    if (s.bias && s.residual) return false;   // as for the 256-row kernels: the residual variants carry no bias path
    tiles += rn_cdiv((long long)s.N * s.Ho * s.Wo, 512) * rn_cdiv(cp, 128);
  }
  if (!conv_halo_shape(p, 512)) return false;
  // enough tiles to fill the chip once (opts.conv_tile = 2 forces the form: tests at small sizes)
  return p->opts.conv_tile >= 2 || tiles >= (p->opts.conv_big_min_tiles > 0 ? p->opts.conv_big_min_tiles : 128);
}

// Which kernel a problem runs on (rn_conv_kernel_id): 0 = 128-row conv_fwd_kernel, 1 = conv_big_kernel, 2 = conv_halo_kernel
// with 256 x 256 tiles, 3 = conv_halo_kernel with 512 x 128 tiles.
// A 3x3 / stride 1 launch that qualifies for the 256 x 256 halo tiles runs as 512 x 128 tiles instead when its channel
// count is a multiple of 128 and the longer patches fit: the same number of tiles and MACs per tile, but a K chunk stages
// ~117 KB instead of ~171 KB per workgroup (one 8 KB weight piece per tap instead of 16 KB; the patch of 512 consecutive
// pixels has relatively fewer halo rows).  Measured inside the step on one box (round 4): head-tower launches 564 -> 511 us,
// class prediction 1460 -> 1333, ResNet stage-3 3x3 (200 tiles) 59.8 -> 56.8, batch-8 towers 157 -> 152.  conv_tile = 2
// keeps the 256 x 256 form (tests, A/B), and so do the small launches that only run here because a split-K workspace lets
// every tile be cut along K.
static int conv_pick(const rn_conv_problem* p) {
  if (conv_use_big(p)) {
    if (!conv_halo_shape(p, 256)) return 1;
    if (p->opts.conv_tile == 0 && p->opts.conv_big_min_tiles == 0) {
      // (a launch of fewer tiles than the 256-row kernels normally take got here through its split-K workspace: every
      // tile cut along K on the 256 x 256 form)
      long long tiles256 = 0;
      for (int i = 0; i < p->num_segments; ++i)
        tiles256 += rn_cdiv((long long)p->seg[i].N * p->seg[i].Ho * p->seg[i].Wo, 256) * rn_cdiv(seg_cols(p->seg[i]), 256);
      if (tiles256 < 192) return 2;
      rn_conv_problem q = *p;
      q.opts.conv_tile = 3;
      if (conv_use_halo512(&q)) return 3;
    }
    return 2;
  }
  return conv_use_halo512(p) ? 3 : 0;
}

// K step of the 128-row kernel: 64, or 32 when the padded channel count is not a multiple of 64 — and for the shallow
// layers (K = R S Cin <= 256: ResNet stage 1's 256 -> 64, the first 1x1 of stage 2), which are HBM-bound: four stages of half
// the size stream better than two or three (tools/probes/ab_conv128_bk.sh: 256 -> 64 at 160 x 160, batch 32, 133.9 -> 119.3 us;
// 256 -> 128 194.0 -> 183.0) while every deeper layer loses 10 - 15 % to the second barrier per 16 MFMAs.
// RNET_CONV128_BK=32 / 64 (A/B probe) forces one of them where the channel count allows.
static int conv128_bk(const rn_conv_problem* p) {
  static const int forced = getenv("RNET_CONV128_BK") ? atoi(getenv("RNET_CONV128_BK")) : 0;
  const int cin = rn_conv_cin_pad(p->seg[0].Cin);
  if (cin % 64 != 0 || forced == 32) return 32;
  if (forced == 64) return 64;
  return (long long)p->R * p->S * cin <= 256 ? 32 : 64;
}
// Tile shape the 128-row kernel runs a problem with: BN = 64 for Cout <= 64 and for small launches (see
// rn_conv2d_nhwc_fwd), BK = 64 unless the padded channel count is not a multiple of 64; returns the tile count.
static int conv128_shape(const rn_conv_problem* p, int* BN_out, int* BK_out) {
  int BN = rn_conv_cout_pad(seg_cols(p->seg[0])) <= 64 ? 64 : 128;
  const int BK = conv128_bk(p);
  if (BN == 128) {
    long long t128 = 0;
    for (int i = 0; i < p->num_segments; ++i)
      t128 += rn_cdiv((long long)p->seg[i].N * p->seg[i].Ho * p->seg[i].Wo, 128) * rn_cdiv(rn_conv_cout_pad(seg_cols(p->seg[i])), 128);
    if (2 * t128 <= rn_num_cus() && !p->opts.ablate && p->opts.conv_tile != 1) BN = 64;   // conv_tile = 1 keeps 128 x 128 (tests)
  }
  long long tiles = 0;
  for (int i = 0; i < p->num_segments; ++i)
    tiles += rn_cdiv((long long)p->seg[i].N * p->seg[i].Ho * p->seg[i].Wo, 128) * rn_cdiv(rn_conv_cout_pad(seg_cols(p->seg[i])), BN);
  *BN_out = BN; *BK_out = BK;
  return tiles > 0x7fffffff ? 0x7fffffff : (int)tiles;
}
// Parts every tile of a 128-row launch is cut into along K (1: whole tiles).  Enough parts to put about one workgroup on
// three of every four compute units (opts.splitk_target_blocks moves the target), at least RN_SPLIT128_MIN_STEPS K steps per part (a
// part costs a 32 - 64 KB partial tile written and read back and a ~2 us hand-off), at most 8 parts (the last arriver
// keeps one 16-byte load per part in flight), the slots must fit the workspace and the tiles its 4096 counters.
#define RN_SPLIT128_MIN_STEPS 4
static int conv128_split_parts(const rn_conv_problem* p, int tiles, int BN, int BK) {
  if (!p->splitk_ws || p->opts.ablate || p->opts.conv_tile == 1 || tiles < 1 || tiles > 4096) return 1;
  int ksteps = 0x7fffffff;   // of the launch's shallowest segment
  for (int i = 0; i < p->num_segments; ++i) {
    const int terms = p->seg[i].w_terms > 1 ? p->seg[i].w_terms : 1;
    const int ks = p->R * p->S * (terms * rn_conv_cin_pad(p->seg[i].Cin) / BK);
    ksteps = ks < ksteps ? ks : ksteps;
  }
  // default target: three quarters of the compute units.  Same-box sweep with the counted-wait K loop, three rounds
  // (tools/bench_infer.py --split-target): batch-1 serving 1.357 / 1.360 / 1.353 ms at 256 workgroups, 1.322 / 1.310 / 1.318 at
  // 192, 1.323 / 1.323 / 1.315 at 160; batch 8 within +-0.4 % of each other (fewer, longer parts: less exchange traffic).
  const int target = p->opts.splitk_target_blocks > 0 ? p->opts.splitk_target_blocks : rn_num_cus() * 3 / 4;
  int S = target / tiles;
  if (S > ksteps / RN_SPLIT128_MIN_STEPS) S = ksteps / RN_SPLIT128_MIN_STEPS;
  if (S > 8) S = 8;
  const long long slot = 128ll * BN * 4;
  while (S >= 2 && RN_SPLITK_HEADER_BYTES + (long long)tiles * S * slot > p->splitk_ws_bytes) --S;
  return S >= 2 ? S : 1;
}

// Balanced tiles for conv_big_kernel's HBM-bound 1x1 launches.  The persistent grid walks its 256-row tiles in rounds of one
// per workgroup; a launch of 3.1 rounds runs as 4 with most of the chip idle in the last one (ResNet stage 3 `*_out` at
// B = 32: 800 tiles on 256 workgroups), and a tile's time there is set by its bytes, not by its MFMAs (four to sixteen K
// steps between a pipeline refill and a 128 KB epilogue).  A 1x1 tile's rows are independent, so the SAME number of rounds
// can be cut finer: rows = ceil(M / floor(rounds * grid / column tiles)) pixels per tile instead of 256 — every workgroup
// then walks `rounds` tiles of rows/256 of the bytes each (the rows a tile does not cover are masked: zero-filled by the
// DMA, never stored; their MFMAs run on zeros).  The busiest workgroup of the 256-row plan keeps its tile count and moves
// fewer bytes; nobody moves more.  Single-segment 1x1 / stride 1 launches only, shallow enough that bytes set the pace
// (K <= 512), and only when it shortens the tiles by 8 % or more; opts.conv_tile / conv_big_min_tiles / max_workgroups
// (tests, A/B) keep whole tiles.  Results are the same values: a tile's accumulation order does not depend on its rows.
// Returns the rows per tile, 0 = whole 256-row tiles.  The fused BatchNorm partial sums are still written per (tile,
// half): rn_conv_bn_row_blocks() tells how many 128-row blocks a segment writes.
static int conv_big_balanced_rows(const rn_conv_problem* p) {
  if (p->num_segments != 1 || p->R != 1 || p->S != 1 || p->stride_h != 1 || p->stride_w != 1) return 0;
  if (p->opts.conv_tile || p->opts.conv_big_min_tiles || p->opts.max_workgroups || p->splitk_ws) return 0;
  const rn_conv_segment& s = p->seg[0];
  const int terms = s.w_terms > 1 ? s.w_terms : 1;
  if (terms * rn_conv_cin_pad(s.Cin) > 512) return 0;
  const long long M = (long long)s.N * s.Ho * s.Wo;
  const int n_tiles = (int)rn_cdiv(rn_conv_cout_pad(seg_cols(s)), 256);
  const int G = rn_persistent_grid(0x7fffffff, rn_num_cus(), p->opts);
  const long long T = rn_cdiv(M, 256) * n_tiles;
  if (T <= 0 || G <= 0) return 0;
  const long long rounds = rn_cdiv(T, G);
  const long long m_tiles = rounds * G / n_tiles;   // row blocks that fit `rounds` rounds
  if (m_tiles < 1) return 0;
  long long rows = rn_cdiv(M, m_tiles);
  rows = (rows + 3) / 4 * 4;
  if (rows < 64 || rows > 236) return 0;            // (236 = 0.92 * 256)
  return (int)rows;
}

int rn_splitk_plan(ConvArgs& a, int min_chunks, void* ws, long long ws_bytes, const rn_launch_opts& opts) {
  const int total = a.total_tiles;
  a.split_f = total; a.split_s = 1; a.vtotal = total; a.pad2_ = 0; a.ws = nullptr;
  const int G0 = rn_persistent_grid(0x7fffffff, rn_num_cus(), opts);
  const int L = total % G0;
  int S = ws ? splitk_parts(total, min_chunks, G0, nullptr) : 1;
  while (S >= 2 && RN_SPLITK_HEADER_BYTES + (long long)L * S * RN_SPLITK_SLOT_BYTES > ws_bytes) --S;
  if (S < 2) return total < G0 ? total : G0;
  a.split_f = total - L; a.split_s = S; a.vtotal = L * S; a.ws = (float*)ws;   // vtotal: units of the SPLIT launch
  return a.split_f ? G0 : L * S;
}

// K chunks (32 input channels x all taps) of the launch's shortest tile
static int conv_min_chunks(const rn_conv_problem* p) {
  int mc = 0x7fffffff;
  for (int i = 0; i < p->num_segments; ++i) {
    const int terms = p->seg[i].w_terms > 1 ? p->seg[i].w_terms : 1;
    const int ch = terms * rn_conv_cin_pad(p->seg[i].Cin) / 32;
    mc = ch < mc ? ch : mc;
  }
  return mc;
}
// ... or 0 when this launch cannot split (conv_big_kernel: whole tiles only — its 1x1 layers are HBM-bound)
static int conv_splitk_min_chunks(const rn_conv_problem* p) { return conv_pick(p) == 2 ? conv_min_chunks(p) : 0; }

// what any problem can use on any grid: 16 KB header + 256 accumulator slots (one per part: L * S <= 256 workgroups)
extern "C" size_t rn_conv_splitk_workspace_max_bytes(void) { return RN_SPLITK_HEADER_BYTES + 256ull * RN_SPLITK_SLOT_BYTES; }

extern "C" size_t rn_conv_splitk_workspace_bytes(const rn_conv_problem* p) {
  if (!p || p->num_segments < 1 || p->num_segments > RN_CONV_MAX_SEGMENTS) return 0;
  if (conv_pick(p) == 0) {   // 128-row kernel: every tile cut into S parts of one [128][BN] fp32 slot each
    int BN, BK;
    const int tiles = conv128_shape(p, &BN, &BK);
    rn_conv_problem q = *p;
    if (!q.splitk_ws) { q.splitk_ws = (void*)16; q.splitk_ws_bytes = (long long)rn_conv_splitk_workspace_max_bytes(); }   // "if one were attached"
    const int S = conv128_split_parts(&q, tiles, BN, BK);
    return S >= 2 ? (size_t)(RN_SPLITK_HEADER_BYTES + (long long)tiles * S * 128 * BN * 4) : 0;
  }
  const int mc = conv_splitk_min_chunks(p);
  if (!mc) return 0;
  long long tiles = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_conv_segment& s = p->seg[i];
    tiles += rn_cdiv((long long)s.N * s.Ho * s.Wo, 256) * rn_cdiv(rn_conv_cout_pad(seg_cols(s)), 256);
  }
  long long bytes = 0;
  splitk_parts((int)tiles, mc, rn_persistent_grid(0x7fffffff, rn_num_cus(), p->opts), &bytes);
  return (size_t)bytes;
}

/* 0: 128-row kernel, 1: conv_big_kernel, 2: conv_halo_kernel */
extern "C" int rn_conv_kernel_id(const rn_conv_problem* p) {
  if (!p || p->num_segments < 1 || p->num_segments > RN_CONV_MAX_SEGMENTS) return -1;
  return conv_pick(p);
}

extern "C" int rn_conv_bn_row_blocks(const rn_conv_problem* p, int segment) {
  if (!p || p->num_segments < 1 || p->num_segments > RN_CONV_MAX_SEGMENTS || segment < 0 || segment >= p->num_segments) return 0;
  const int kid = conv_pick(p);
  const long long M = (long long)p->seg[segment].N * p->seg[segment].Ho * p->seg[segment].Wo;
  if (kid == 1) {
    const int r = conv_big_balanced_rows(p);
    if (r) return (int)(2 * rn_cdiv(M, r));
  }
  const int rows = kid == 3 ? 512 : (kid ? 256 : 128);
  return (int)((rows / 128) * rn_cdiv(M, rows));
}

extern "C" int rn_conv_tile_rows(const rn_conv_problem* p) {
  if (!p || p->num_segments < 1 || p->num_segments > RN_CONV_MAX_SEGMENTS) return 0;
  const int kid = conv_pick(p);
  return kid == 3 ? 512 : (kid ? 256 : 128);
}

extern "C" int rn_conv2d_nhwc_fwd(const rn_conv_problem* p, void* stream) {
  RN_CHECK_ARG(p != nullptr, "rn_conv2d_nhwc_fwd: null problem");
  RN_CHECK_ARG(p->num_segments >= 1 && p->num_segments <= RN_CONV_MAX_SEGMENTS,
               "rn_conv2d_nhwc_fwd: num_segments=%d", p->num_segments);
  RN_CHECK_ARG(p->R >= 1 && p->S >= 1 && p->R * p->S <= 32, "rn_conv2d_nhwc_fwd: R*S=%d > 32", p->R * p->S);
  RN_CHECK_ARG(p->stride_h >= 1 && p->stride_w >= 1, "rn_conv2d_nhwc_fwd: bad stride");
  RN_CHECK_ARG(p->out_dtype == RN_DT_BF16 || p->out_dtype == RN_DT_F32, "rn_conv2d_nhwc_fwd: bad out_dtype");
  if (const int orc = rn_validate_launch_opts(p->opts, "rn_conv2d_nhwc_fwd")) return orc;
  ConvArgs a;
  a.R = p->R; a.S = p->S; a.sh = p->stride_h; a.sw = p->stride_w; a.pt = p->pad_top; a.pl = p->pad_left;
  a.act = p->act; a.nseg = p->num_segments; a.pad_ = 0;
  const int cout_pad0 = rn_conv_cout_pad(seg_cols(p->seg[0]));
  int BN = cout_pad0 <= 64 ? 64 : 128;
  // K step: 64 unless the (padded) channel count is small; Cin need only be a multiple of 8 — the
  // tail of the last K step reads past the pixel's channels (or out of range -> zeros) and meets the
  // zero-padded weight columns, so it contributes nothing.
  const int BK = conv128_bk(p);
  const int kid = conv_pick(p);
  const bool big = kid == 1 || kid == 2;
  const bool halo512 = kid == 3;   // 512 x 128 tiles of the halo kernel
  const int BM = big ? 256 : (halo512 ? 512 : 128);
  if (!big && !halo512) {
    // Small launches (batch-8 inference, ResNet stage 4: 100 tiles of 128 x 128 on 256 CUs): 128 x 64 tiles put the work
    // on twice as many CUs and read 12 KB instead of 16 KB of LDS fragments per wave and K step — the 128-row kernel is
    // bound by fragment bandwidth at one workgroup per CU (DESIGN.md section 4, round-3 probes).  Only while the
    // narrower tiles still fit one per CU: at two per CU they share that bandwidth again.  (conv128_shape)
    int bk_;
    conv128_shape(p, &BN, &bk_);
  }
  const int BNT = big ? 256 : BN;   // n-tile width
  const int bal_rows = kid == 1 ? conv_big_balanced_rows(p) : 0;
  int tiles = 0;
  // conv_big_kernel launch whose segments are all one column tile wide but differ 2x or more in K depth (the FPN lateral 1x1
  // convs: 512 / 1024 / 2048 input channels): tiles numbered deepest segment first and dealt to the workgroups round-robin
  // (identity numbering) instead of in the XCD-contiguous ranges that keep neighbouring column tiles on one L2 — there are
  // no neighbouring column tiles here, and a contiguous range hands one XCD all of the 64-step tiles (181 K steps per CU
  // there against 87 on average).  The order is internal to the launch: every tile's result is what it was.
  int order[RN_CONV_MAX_SEGMENTS];
  for (int i = 0; i < p->num_segments; ++i) order[i] = i;
  bool deal = false;
  if (kid == 1 && p->num_segments > 1) {
    long long dmin = 1ll << 60, dmax = 0;
    bool one_col = true;
    for (int i = 0; i < p->num_segments; ++i) {
      const long long depth = (long long)p->R * p->S * rn_conv_cin_pad(p->seg[i].Cin) * (p->seg[i].w_terms > 1 ? p->seg[i].w_terms : 1);
      dmin = depth < dmin ? depth : dmin;
      dmax = depth > dmax ? depth : dmax;
      one_col = one_col && rn_conv_cout_pad(seg_cols(p->seg[i])) <= 256;
    }
    deal = one_col && dmax >= 2 * dmin;
    if (deal)
      std::stable_sort(order, order + p->num_segments, [&](int x, int y) {
        return rn_conv_cin_pad(p->seg[x].Cin) * (p->seg[x].w_terms > 1 ? p->seg[x].w_terms : 1) >
               rn_conv_cin_pad(p->seg[y].Cin) * (p->seg[y].w_terms > 1 ? p->seg[y].w_terms : 1);
      });
  }
  for (int ii = 0; ii < p->num_segments; ++ii) {
    const int i = order[ii];
    const rn_conv_segment& s = p->seg[i];
    RN_CHECK_ARG(s.x && s.w && s.y, "rn_conv2d_nhwc_fwd: segment %d has a null tensor", i);
    RN_CHECK_ARG(s.N > 0 && s.H > 0 && s.W > 0 && s.Ho > 0 && s.Wo > 0 && s.Cout > 0 && s.Cin > 0,
                 "rn_conv2d_nhwc_fwd: segment %d bad shape", i);
    RN_CHECK_ARG(s.Cin % 8 == 0 && rn_conv_cin_pad(s.Cin) % BK == 0,
                 "rn_conv2d_nhwc_fwd: segment %d Cin=%d must be a multiple of 8 (K step %d)", i, s.Cin, BK);
    RN_CHECK_ARG(s.pix_stride % 4 == 0 && s.pix_stride > 0,
                 "rn_conv2d_nhwc_fwd: segment %d pix_stride=%d must be a positive multiple of 4", i, s.pix_stride);
    RN_CHECK_ARG(s.Cout % 4 == 0, "rn_conv2d_nhwc_fwd: segment %d Cout=%d not a multiple of 4", i, s.Cout);
    const int cp = rn_conv_cout_pad(seg_cols(s));
    RN_CHECK_ARG((cp <= 64) == (cout_pad0 <= 64), "rn_conv2d_nhwc_fwd: segments mix Cout tile widths");
    RN_CHECK_ARG(!s.w_pair || ((big || halo512) && p->out_dtype == RN_DT_F32 && s.w_terms <= 1 && !s.scale && !s.shift &&
                               !s.residual && !s.bn_partial),
                 "rn_conv2d_nhwc_fwd: segment %d: w_pair needs an f32 launch without scale / shift / residual that the 256- / "
                 "512-row kernels take (rn_conv_kernel_id() != 0)", i);
    RN_CHECK_ARG(((uintptr_t)s.x | (uintptr_t)s.w | (uintptr_t)s.y | (uintptr_t)s.residual) % 16 == 0,
                 "rn_conv2d_nhwc_fwd: segment %d tensors must be 16-byte aligned", i);
    const long long M = (long long)s.N * s.Ho * s.Wo;
    RN_CHECK_ARG(M < (1ll << 31) && (long long)s.N * s.H * s.W * s.pix_stride * 2 < (1ll << 31),
                 "rn_conv2d_nhwc_fwd: segment %d input exceeds the 2 GiB buffer-addressing limit", i);
    // the last input row/col a valid tap may touch must be inside the image
    ConvSegDev& d = a.seg[ii];
    d.x = (const uint16_t*)s.x; d.w = (const uint16_t*)s.w; d.y = s.y;
    d.scale = s.scale; d.shift = s.shift; d.residual = (const uint16_t*)s.residual;
    d.bn_partial = s.bn_partial;
    d.bn_y = (const uint16_t*)s.bn_bwd_y;
    d.bn_fwd = s.bn_bwd_fwd;
    if (s.bn_bwd_y) {
      RN_CHECK_ARG(s.bn_partial && s.bn_bwd_fwd && p->out_dtype == RN_DT_BF16 && !s.scale && !s.shift && !s.bias &&
                       !s.residual && p->act == RN_ACT_NONE && s.Cout % 8 == 0 && (uintptr_t)s.bn_bwd_y % 16 == 0,
                   "rn_conv2d_nhwc_fwd: segment %d: bn_bwd_y needs bn_partial + bn_bwd_fwd on a plain bf16 launch", i);
    }
    RN_CHECK_ARG((s.bn_bwd_y != nullptr) == (p->seg[0].bn_bwd_y != nullptr),
                 "rn_conv2d_nhwc_fwd: bn_bwd_y must be set on all segments or none");
    d.bias = s.bias;
    d.N = s.N; d.H = s.H; d.W = s.W; d.Cin = s.Cin; d.pix_stride = s.pix_stride;
    d.Ho = s.Ho; d.Wo = s.Wo; d.Cout = seg_cols(s);
    d.pair_cout = s.w_pair ? s.Cout : 0;
    d.rows = kid == 1 ? bal_rows : 0;   // conv_big_kernel: balanced tiles (single-segment launches only)
    d.M = (int)M;
    d.tile_begin = tiles;
    d.n_tiles = (int)rn_cdiv(cp, BNT);
    const int terms = s.w_terms > 1 ? s.w_terms : 1;
    RN_CHECK_ARG(terms <= 3, "rn_conv2d_nhwc_fwd: segment %d w_terms=%d (1..3)", i, s.w_terms);
    d.cwrap = rn_conv_cin_pad(s.Cin);
    d.CinP = terms * d.cwrap;
    d.halo_pitch = s.W + 1;
    if (rn_conv_halo_patch_pixels(s.N, s.H, s.W, rn_conv_halo_pitch(s.W), halo512 ? 512 : 256) <= rn_conv_halo_capacity(halo512 ? 512 : 256))
      d.halo_pitch = rn_conv_halo_pitch(s.W);
    tiles += (int)rn_cdiv(M, d.rows ? d.rows : BM) * d.n_tiles;
  }
  a.total_tiles = tiles;
  a.split_f = a.vtotal = tiles; a.split_s = 1; a.pad2_ = 0; a.ws = nullptr;
  RN_CHECK_ARG(p->splitk_ws == nullptr || ((uintptr_t)p->splitk_ws % 16 == 0 && p->splitk_ws_bytes >= 0),
               "rn_conv2d_nhwc_fwd: splitk_ws must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = p->out_dtype == RN_DT_F32;
  if (p->opts.ablate && BM == 128 && BN == 128 && BK == 64 && !f32) {
    switch (p->opts.ablate) {
      case 1: return launch_ablate<1>(a, st);
      case 2: return launch_ablate<2>(a, st);
      case 3: return launch_ablate<3>(a, st);
      case 4: return launch_ablate<4>(a, st);
      case 5: return launch_ablate<5>(a, st);
      case 6: return launch_ablate<6>(a, st);
      case 7: return launch_ablate<7>(a, st);
      case 8: return launch_ablate<8>(a, st);
      case 16: return launch_ablate<16>(a, st);
      case 32: return launch_ablate<32>(a, st);
      case 48: return launch_ablate<48>(a, st);
      default: break;
    }
  }
  if (halo512) return rn_launch_conv_halo(a, f32, p->opts, st, 4);
  if (kid == 2) {
    rn_splitk_plan(a, conv_splitk_min_chunks(p), p->splitk_ws, p->splitk_ws_bytes, p->opts);
    return rn_launch_conv_halo(a, f32, p->opts, st);
  }
  if (big) {
    a.pad_ = 1;   // bit 0: float-reciprocal index arithmetic in the tile set-up, valid while every M < 2^22
    for (int i = 0; i < a.nseg; ++i)
      if (a.seg[i].M >= (1 << 22)) a.pad_ = 0;
    if (deal) a.pad_ |= 2;   // bit 1: tiles dealt round-robin (see above)
    return rn_launch_conv_big(a, f32, p->opts, st);
  }
  a.pad_ = 1;   // bit 0: float-reciprocal index arithmetic in the prologue, valid while every M < 2^22
  for (int i = 0; i < a.nseg; ++i)
    if (a.seg[i].M >= (1 << 22)) a.pad_ = 0;
  // 128-row kernel.  With a split-K workspace a SMALL launch of a deep layer (fewer tiles than the chip has compute units:
  // batch-1 / batch-8 inference, ResNet stage 3 / 4, the FPN laterals) cuts every tile along K (conv128_split_parts).
  const int S128 = conv128_split_parts(p, tiles, BN, BK);
  if (S128 >= 2) {
    a.split_s = S128; a.vtotal = tiles * S128; a.ws = (float*)p->splitk_ws;
    return launch_conv128<true>(a, BN, BK, f32, st);
  }
  return launch_conv128<false>(a, BN, BK, f32, st);
}

// ---- weight / input packing ----------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w, int R, int S, int Cin, int Cout, int Cin_pad,
                                   int Cout_pad, uint16_t* __restrict__ out) {
  const long long total = (long long)Cout_pad * R * S * Cin_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin_pad);
    long long t = i / Cin_pad;
    const int s = (int)(t % S);
    t /= S;
    const int r = (int)(t % R);
    const int o = (int)(t / R);
    float v = 0.0f;
    if (o < Cout && c < Cin) v = w[(((long long)r * S + s) * Cin + c) * Cout + o];
    out[i] = rn_f32_to_bf16(v);
  }
}

extern "C" int rn_pack_conv_weight(const float* w_hwio, int R, int S, int Cin, int Cout, int Cin_pad,
                                   void* w_packed, void* stream) {
  RN_CHECK_ARG(w_hwio && w_packed && R > 0 && S > 0 && Cin > 0 && Cout > 0 && Cin_pad >= Cin,
               "rn_pack_conv_weight: bad argument");
  const int Cout_pad = rn_conv_cout_pad(Cout);
  const long long total = (long long)Cout_pad * R * S * Cin_pad;
  int blocks = (int)(rn_cdiv(total, 256) < 4096 ? rn_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, R, S, Cin,
                     Cout, Cin_pad, Cout_pad, (uint16_t*)w_packed);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// the same packing from the training engine's master layout [Cout][R][S][Cin] (f32)
__global__ void pack_weight_ohwi_kernel(const float* __restrict__ w, int RS, int Cin, int Cout, int Cin_pad,
                                        int Cout_pad, uint16_t* __restrict__ out) {
  const long long total = (long long)Cout_pad * RS * Cin_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin_pad);
    const long long t = i / Cin_pad;   // = o * RS + tap
    float v = 0.0f;
    if (t < (long long)Cout * RS && c < Cin) v = w[t * Cin + c];
    out[i] = rn_f32_to_bf16(v);
  }
}

extern "C" int rn_pack_conv_weight_ohwi(const float* w_ohwi, int R, int S, int Cin, int Cout, int Cin_pad,
                                        void* w_packed, void* stream) {
  RN_CHECK_ARG(w_ohwi && w_packed && R > 0 && S > 0 && Cin > 0 && Cout > 0 && Cin_pad >= Cin,
               "rn_pack_conv_weight_ohwi: bad argument");
  const int Cout_pad = rn_conv_cout_pad(Cout);
  const long long total = (long long)Cout_pad * R * S * Cin_pad;
  int blocks = (int)(rn_cdiv(total, 256) < 4096 ? rn_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(pack_weight_ohwi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ohwi, R * S, Cin,
                     Cout, Cin_pad, Cout_pad, (uint16_t*)w_packed);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// split-bf16 planes of an f32 kernel (rn_conv_segment.w_terms): [Cout_pad][R*S][terms][Cin_pad], plane 0 = rb(w),
// plane t = rb(w - planes before it) — x (a bf16 tensor) times the f32 weight, to 16 / 24 weight mantissa bits
__global__ void pack_weight_split_kernel(const float* __restrict__ w, int ohwi, int RS, int Cin, int Cout, int Cin_pad,
                                         int Cout_pad, int terms, uint16_t* __restrict__ out) {
  const long long total = (long long)Cout_pad * RS * Cin_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin_pad);
    const long long t = i / Cin_pad;   // = o * RS + tap
    const int tap = (int)(t % RS), o = (int)(t / RS);
    float v = 0.0f;
    if (o < Cout && c < Cin) v = ohwi ? w[t * Cin + c] : w[((long long)tap * Cin + c) * Cout + o];
    uint16_t* dst = out + (t * terms) * Cin_pad + c;
    for (int k = 0; k < terms; ++k) {
      const uint16_t b = rn_f32_to_bf16(v);
      dst[(long long)k * Cin_pad] = b;
      v -= rn_bf16_to_f32(b);   // exact: the residual of a round-to-nearest is representable in fp32
    }
  }
}

// rn_conv_segment.w_pair: the two planes along Cout — packed row 64b + 32h + c = plane h of channel 32b + c
__global__ void pack_weight_pair_kernel(const float* __restrict__ w, int ohwi, int RS, int Cin, int Cout, int Cin_pad,
                                        int rows, uint16_t* __restrict__ out) {
  const long long total = (long long)rows * RS * Cin_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin_pad);
    const long long t = i / Cin_pad;   // = row * RS + tap
    const int tap = (int)(t % RS), row = (int)(t / RS);
    const int o = (row >> 6) * 32 + (row & 31), plane = (row >> 5) & 1;
    float v = 0.0f;
    if (o < Cout && c < Cin) v = ohwi ? w[((long long)o * RS + tap) * Cin + c] : w[((long long)tap * Cin + c) * Cout + o];
    const uint16_t b = rn_f32_to_bf16(v);
    out[i] = plane ? rn_f32_to_bf16(v - rn_bf16_to_f32(b)) : b;
  }
}

extern "C" int rn_pack_conv_weight_pair(const float* w, int layout_ohwi, int R, int S, int Cin, int Cout, int Cin_pad,
                                        void* w_packed, void* stream) {
  RN_CHECK_ARG(w && w_packed && R > 0 && S > 0 && Cin > 0 && Cout > 0 && Cin_pad >= Cin,
               "rn_pack_conv_weight_pair: bad argument");
  const int rows = rn_conv_pair_rows(Cout);
  const long long total = (long long)rows * R * S * Cin_pad;
  int blocks = (int)(rn_cdiv(total, 256) < 4096 ? rn_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(pack_weight_pair_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, layout_ohwi, R * S,
                     Cin, Cout, Cin_pad, rows, (uint16_t*)w_packed);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_pack_conv_weight_split(const float* w, int layout_ohwi, int R, int S, int Cin, int Cout, int Cin_pad,
                                         int terms, void* w_packed, void* stream) {
  RN_CHECK_ARG(w && w_packed && R > 0 && S > 0 && Cin > 0 && Cout > 0 && Cin_pad >= Cin && terms >= 1 && terms <= 3,
               "rn_pack_conv_weight_split: bad argument");
  const int Cout_pad = rn_conv_cout_pad(Cout);
  const long long total = (long long)Cout_pad * R * S * Cin_pad;
  int blocks = (int)(rn_cdiv(total, 256) < 4096 ? rn_cdiv(total, 256) : 4096);
  hipLaunchKernelGGL(pack_weight_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, layout_ohwi, R * S,
                     Cin, Cout, Cin_pad, Cout_pad, terms, (uint16_t*)w_packed);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// stem: HWIO [R,S,3,Cout] (S <= 8) -> [Cout_pad][R][8 taps][4 ch]; taps >= S and channel 3 are zero
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, int R, int S, int Cout, int Cout_pad,
                                        uint16_t* __restrict__ out) {
  const int total = Cout_pad * R * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int c = i & 3, s = (i >> 2) & 7, r = (i >> 5) % R, o = i / (R * 32);
    float v = 0.0f;
    if (o < Cout && c < 3 && s < S) v = w[((r * S + s) * 3 + c) * Cout + o];
    out[i] = rn_f32_to_bf16(v);
  }
}

extern "C" int rn_pack_stem_weight_rs(const float* w_hwio, int R, int S, int Cout, void* w_packed, void* stream) {
  RN_CHECK_ARG(w_hwio && w_packed && Cout > 0 && R > 0 && S > 0 && S <= 8, "rn_pack_stem_weight_rs: bad argument");
  const int Cout_pad = rn_conv_cout_pad(Cout);
  hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((Cout_pad * R * 32 + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, w_hwio, R, S, Cout, Cout_pad, (uint16_t*)w_packed);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_pack_stem_weight(const float* w_hwio, int Cout, void* w_packed, void* stream) {
  return rn_pack_stem_weight_rs(w_hwio, 7, 7, Cout, w_packed, stream);
}

extern "C" int rn_stem_padded_width(int W) { return (int)rn_align_up((size_t)W + 6 + 2, 8); }

// images f32 [N,H,W,3] -> bf16 [N,H+6,Wp,4], zero border of 3 (fixed_padding for k=7) and zero
// 4th channel.  One thread per output pixel (8-byte store); reads are 12-byte pixels.
__global__ void __launch_bounds__(256)
pack_stem_input_kernel(const float* __restrict__ img, int N, int H, int W, int pad_top, int pad_left, int Hp,
                       int Wp, uint2* __restrict__ out) {
  const long long total = (long long)N * Hp * Wp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp);
    const long long t = i / Wp;
    const int yp = (int)(t % Hp);
    const int n = (int)(t / Hp);
    const int x = xp - pad_left, y = yp - pad_top;
    uint2 o = make_uint2(0u, 0u);
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
      const float* p = img + (((long long)n * H + y) * W + x) * 3;
      o.x = rn_pack_bf16x2(p[0], p[1]);
      o.y = rn_pack_bf16x2(p[2], 0.0f);
    }
    out[i] = o;
  }
}

extern "C" int rn_pack_image_nhwc4(const float* images, int N, int H, int W, int pad_top, int pad_left, int Hp,
                                   int Wp, void* packed, void* stream) {
  RN_CHECK_ARG(images && packed && N > 0 && H > 0 && W > 0 && pad_top >= 0 && pad_left >= 0 &&
                   Hp >= H + pad_top && Wp >= W + pad_left && Wp % 8 == 0,
               "rn_pack_image_nhwc4: bad argument");
  const long long total = (long long)N * Hp * Wp;
  int blocks = (int)(rn_cdiv(total, 256) < 8192 ? rn_cdiv(total, 256) : 8192);
  hipLaunchKernelGGL(pack_stem_input_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, images, N, H, W,
                     pad_top, pad_left, Hp, Wp, (uint2*)packed);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_pack_stem_input(const float* images, int N, int H, int W, void* packed, void* stream) {
  RN_CHECK_ARG(images && packed && N > 0 && H > 0 && W > 0, "rn_pack_stem_input: bad argument");
  return rn_pack_image_nhwc4(images, N, H, W, 3, 3, H + 6, rn_stem_padded_width(W), packed, stream);
}
