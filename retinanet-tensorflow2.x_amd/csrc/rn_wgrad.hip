// rn_wgrad.hip — weight gradient of the NHWC convolutions (backward of K1/K2; the autodiff of
// tf.keras.layers.Conv2D at resnet.py:137-144, fpn.py:47-66, detection_head.py:56-88 taken by
// tape.gradient in executor.py:427-428).
//
//   dW[co][r][s][ci] = sum over output pixels p of dy[p][co] * x[pixel(p) shifted by tap (r,s)][ci]
//
// GEMM view per filter tap: M = Cout, N = Cin, K = output pixels (all images, and for the shared
// head convs all pyramid levels).  Both operands are pixel-major ([pixel][channel], the
// reduction index is the SLOW axis), so neither has the k-contiguous fragment MFMA wants; tiles
// are DMA'd into LDS as they lie in HBM and the fragments are read with the gfx950 transpose
// read ds_read_b64_tr_b16 (each 16-lane group turns a 4-pixel x 16-channel block into
// 4 k-values per lane).  The tap shift is then just a row (pixel) offset of the x tile.
//   * workgroup = 256 threads, C tile 128 (co) x 128 (ci) for ONE tap, K step 64 pixels,
//     2 LDS stages fed by buffer_load...lds (OOB rows -> zeros: image borders, chunk tails);
//   * LDS rows are 256 B (128 channels); the 16-byte slot is XOR-swizzled by (row & 3) << 2 so
//     the 4 pixel rows a transpose read touches fall in different bank groups;
//   * split-K over pixel chunks for parallelism: every (tile, chunk) workgroup writes its fp32
//     partial tile to the workspace, a second kernel adds the chunks in index order
//     (deterministic; no float atomics).
// MFMA-bound for large layers; the partial-tile traffic is chunks * |W| * 4 B.
#include <string.h>

#include "rn_wgrad_dev.h"

__device__ __forceinline__ void wg_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

// transpose-read fragment: 8 k-values (pixels kb..kb+7 of this lane's k half) for channel `col`
__device__ __forceinline__ bf16x8_t wg_frag(const char* tile, int kb, int lane, int col0) {
  const int g = lane >> 4, i = lane & 15;
  const int col = col0 + 16 * (g & 1) + 4 * (i & 3);
  const int row0 = kb + (g >> 1) * 8 + (i >> 2);
  const int slot = col >> 3, sub = (col & 7) * 2;
  const int r0 = row0, r1 = row0 + 4;
  const lds_b4_t* p0 = (const lds_b4_t*)(tile + r0 * 256 + ((slot ^ ((r0 & 3) << 2)) << 4) + sub);
  const lds_b4_t* p1 = (const lds_b4_t*)(tile + r1 * 256 + ((slot ^ ((r1 & 3) << 2)) << 4) + sub);
  const bf16x4_t lo = rn_ds_read_tr4(p0);
  const bf16x4_t hi = rn_ds_read_tr4(p1);
  bf16x8_t r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// LINEAR: see wgrad_big_kernel — stride 1, same-size, symmetric padding: constant source-offset advance per K step
template <bool LINEAR>
__global__ void __launch_bounds__(WG_THREADS, 2) wgrad_kernel(const WgArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // dy: 2 stages x 16 KB, then x: 3 stages x 16 KB
  // XCD-aware block -> (chunk, tile) map: blocks are dispatched round-robin over the 8 XCDs, so give
  // each XCD a contiguous range of logical ids with the tile index fastest — every (co, ci, tap)
  // tile of one pixel chunk then runs on the same XCD and re-reads that chunk's x / dy rows from its
  // L2 instead of HBM (first version: 2.7 TB/s of fetch traffic, each chunk pulled by all 8 XCDs).
  // A pixel chunk's tiles only share L2 lines while they walk the chunk in lockstep, i.e. while they are
  // all resident at once: an XCD holds 64 workgroups, so launches with more tiles per chunk (class prediction
  // conv: 6 x 2 x 9 = 108) go through the chunk in `co_groups` passes of <= 64 tiles (464 -> ~900 TFLOP/s).
  const int tiles_per_tap = args.gco * args.ci_tiles;
  const int tiles_all = tiles_per_tap * args.R * args.S;
  int logical;
  {
    const int total = tiles_all * args.total_chunks * args.co_groups;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, rr = total & 7;
    logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  }
  const int vchunk = logical / tiles_all;
  const int chunk = vchunk / args.co_groups, cgp = vchunk - chunk * args.co_groups;
  const int tile_id = logical - vchunk * tiles_all;
  const int tap = tile_id / tiles_per_tap;
  const int tt = tile_id - tap * tiles_per_tap;
  const int co_t = cgp * args.gco + tt / args.ci_tiles, ci_t = tt % args.ci_tiles;
  if (co_t >= args.co_tiles) return;   // padding of the last group (whole workgroup, before any barrier)
  const int co0 = co_t * 128, ci0 = ci_t * 128;
  const int r = tap / args.S, s = tap - r * args.S;
  int si = 0;
#pragma unroll 1
  for (int i = 1; i < args.nseg; ++i)
    if (chunk >= args.seg[i].chunk_begin) si = i;
  const WgSegDev& sg = args.seg[si];
  const int p_begin = (chunk - sg.chunk_begin) * args.CH;
  const int p_end = (p_begin + args.CH) < sg.P ? (p_begin + args.CH) : sg.P;
  const int Cin = args.Cin, Cout = args.Cout;
  const int H = sg.H, W = sg.W, Ho = sg.Ho, Wo = sg.Wo;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 1, wave_n = wave & 1;

  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.dy, 0, (int)((long long)sg.P * sg.dyS * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.x, 0, (int)((long long)sg.N * H * W * sg.xS * 2), 0x00020000);

  // DMA bookkeeping: instruction j of this wave fills rows (j*4 + wave)*4 .. +3 of a 64-row tile
  const int d_row = lane >> 4, d_pos = lane & 15;
  const float inv_wo = 1.0f / (float)Wo, inv_ho = 1.0f / (float)Ho;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

  const int ksteps = (p_end - p_begin + WG_BK - 1) / WG_BK;

  // (n, oy, ox) of this lane's 4 DMA rows: decomposed once (float reciprocal + one correction, p < 2^24),
  // then advanced by the K step with adds and compares only — the first version redid the two divisions
  // for every row of every K step, ~600 VALU cycles per wave per K step against 512 cycles of MFMA.
  int r_ox[4], r_oy[4], r_n[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p0 = p_begin + (j * 4 + wave) * 4 + d_row;
    int t2 = (int)((float)p0 * inv_wo);
    int ox = p0 - t2 * Wo;
    if (ox < 0) { ox += Wo; --t2; } else if (ox >= Wo) { ox -= Wo; ++t2; }
    int n = (int)((float)t2 * inv_ho);
    int oy = t2 - n * Ho;
    if (oy < 0) { oy += Ho; --n; } else if (oy >= Ho) { oy -= Ho; ++n; }
    r_ox[j] = ox; r_oy[j] = oy; r_n[j] = n;
  }
  const int adv_q = WG_BK / Wo, adv_r = WG_BK - adv_q * Wo;      // 64 pixels = adv_q rows + adv_r columns
  const int adv_qn = adv_q / Ho, adv_qr = adv_q - adv_qn * Ho;   //            = adv_qn images + adv_qr rows + ...

  unsigned l_dy[4], l_x[4];
  const int y_bad = r < args.pt ? 0 : (r > args.pt ? Ho - 1 : -1), x_bad = s < args.pl ? 0 : (s > args.pl ? Wo - 1 : -1);
  if (LINEAR) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (j * 4 + wave) * 4 + d_row;
      const int chunkc = d_pos ^ ((row & 3) << 2);
      const int p0 = p_begin + row;
      l_dy[j] = (unsigned)((p0 * sg.dyS + co0 + chunkc * 8) * 2);
      l_x[j] = (unsigned)(((p0 + (r - args.pt) * W + (s - args.pl)) * sg.xS + ci0 + chunkc * 8) * 2);
    }
  }
  const unsigned l_dy_step = (unsigned)(WG_BK * sg.dyS * 2), l_x_step = (unsigned)(WG_BK * sg.xS * 2);

// The two operands ride in separate LDS rings (see the K loop): dy in two stages at smem, x in three behind them.
#define WG_ISSUE_DY(buf, p0_)                                                                     \
  do {                                                                                            \
    char* st__ = smem + (buf) * WG_TILE_BYTES;                                                    \
    if (LINEAR) {                                                                                 \
      const int left__ = p_end - (p0_);                                                           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
        const bool in__ = (j * 4 + wave) * 4 + d_row < left__;                                    \
        wg_dma16(rs_dy, st__ + (j * 4 + wave) * 1024, in__ ? l_dy[j] : WG_OOB);                   \
        l_dy[j] += l_dy_step;                                                                     \
      }                                                                                           \
    } else                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
      const int row__ = (j * 4 + wave) * 4 + d_row;                                               \
      const int chunk__ = d_pos ^ ((row__ & 3) << 2);                                             \
      const int p__ = (p0_) + row__;                                                              \
      const bool in__ = p__ < p_end;                                                              \
      const unsigned va__ = in__ ? (__umul24((unsigned)p__, (unsigned)sg.dyS) + co0 + chunk__ * 8) * 2u : WG_OOB; \
      wg_dma16(rs_dy, st__ + (j * 4 + wave) * 1024, va__);                                        \
    }                                                                                             \
  } while (0)
// (the (n, oy, ox) trackers belong to the x stream: they advance with every x issue)
#define WG_ISSUE_X(buf, p0_)                                                                      \
  do {                                                                                            \
    char* st__ = smem + 2 * WG_TILE_BYTES + (buf) * WG_TILE_BYTES;                                \
    if (LINEAR) {                                                                                 \
      const int left__ = p_end - (p0_);                                                           \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                             \
        const bool in__ = (j * 4 + wave) * 4 + d_row < left__;                                    \
        const bool ok__ = in__ && r_oy[j] != y_bad && r_ox[j] != x_bad;                           \
        wg_dma16(rs_x, st__ + (j * 4 + wave) * 1024, ok__ ? l_x[j] : WG_OOB);                     \
        l_x[j] += l_x_step;                                                                       \
        int ox__ = r_ox[j] + adv_r, oy__ = r_oy[j] + adv_qr;                                      \
        const int c1__ = ox__ >= Wo;                                                              \
        ox__ -= c1__ ? Wo : 0; oy__ += c1__;                                                      \
        oy__ -= oy__ >= Ho ? Ho : 0;                                                              \
        r_ox[j] = ox__; r_oy[j] = oy__;                                                           \
      }                                                                                           \
    } else                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
      const int row__ = (j * 4 + wave) * 4 + d_row;                                               \
      const int chunk__ = d_pos ^ ((row__ & 3) << 2);                                             \
      const int p__ = (p0_) + row__;                                                              \
      const bool in__ = p__ < p_end;                                                              \
      const int iy__ = r_oy[j] * args.sh - args.pt + r, ix__ = r_ox[j] * args.sw - args.pl + s;   \
      const bool ok__ = in__ && (unsigned)iy__ < (unsigned)H && (unsigned)ix__ < (unsigned)W;     \
      /* 24-bit multiplies (full rate): every factor is < 2^24 (host check), the byte offset < 2^31 */ \
      const unsigned pix__ = __umul24(__umul24((unsigned)r_n[j], (unsigned)H) + (unsigned)iy__, (unsigned)W) + (unsigned)ix__; \
      const unsigned vb__ = ok__ ? (__umul24(pix__, (unsigned)sg.xS) + ci0 + chunk__ * 8) * 2u : WG_OOB; \
      wg_dma16(rs_x, st__ + (j * 4 + wave) * 1024, vb__);                                         \
      /* advance this row by one K step */                                                        \
      int ox__ = r_ox[j] + adv_r, oy__ = r_oy[j] + adv_qr, n__ = r_n[j] + adv_qn;                 \
      const int c1__ = ox__ >= Wo;                                                                \
      ox__ -= c1__ ? Wo : 0; oy__ += c1__;                                                        \
      const int c2__ = oy__ >= Ho;                                                                \
      oy__ -= c2__ ? Ho : 0; n__ += c2__;                                                         \
      const int c3__ = oy__ >= Ho;                                                                \
      oy__ -= c3__ ? Ho : 0; n__ += c3__;                                                         \
      r_ox[j] = ox__; r_oy[j] = oy__; r_n[j] = n__;                                               \
    }                                                                                             \
  } while (0)

  // Fragment addresses inside a stage (see wg_frag for the layout): the swizzle term (row & 3) does not depend on the K
  // slice, so one byte offset per 32-channel MFMA tile + immediates (K slice: 16 rows = 4096 B, second half: +1024) cover
  // every transpose read.  The reads are issued from inline asm (rn_wgrad_dev.h): through the builtin the compiler put
  // `s_waitcnt vmcnt(0)` in front of the first read of every K step — i.e. right behind the DMA of the NEXT stage, which
  // was thereby waited for before any MFMA of the current one: no overlap inside a workgroup at all.
  unsigned fbase[4];
  {
    const int g = lane >> 4, i = lane & 15;
    const int row0 = (g >> 1) * 8 + (i >> 2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = (t < 2 ? wave_m : wave_n) * 64 + (t & 1) * 32 + 16 * (g & 1) + 4 * (i & 3);
      fbase[t] = (unsigned)(row0 * 256 + (((col >> 3) ^ ((row0 & 3) << 2)) << 4) + (col & 7) * 2);   // inside its operand's tile
    }
  }
  const unsigned lds0 = rn_lds_addr(smem);
#define WG_READS(q_, st_, kk_)                                                                     \
  do {                                                                                             \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                \
      const unsigned a__ = (t < 2 ? st_##_dy : st_##_x) + fbase[t];                                \
      RN_TR_ISSUE(q_[2 * t], a__, (kk_) * 4096);                                                   \
      RN_TR_ISSUE(q_[2 * t + 1], a__, (kk_) * 4096 + 1024);                                        \
    }                                                                                              \
  } while (0)
#define WG_WAIT(q_, n_)                                                                            \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                             \
               : "+v"(q_[0]), "+v"(q_[1]), "+v"(q_[2]), "+v"(q_[3]), "+v"(q_[4]), "+v"(q_[5]), "+v"(q_[6]), "+v"(q_[7]) \
               : "i"(n_) : "memory")
#define WG_MFMAS(q_)                                                                               \
  do {                                                                                             \
    const bf16x8_t fa0__ = rn_tr_frag(q_[0], q_[1]), fa1__ = rn_tr_frag(q_[2], q_[3]);             \
    const bf16x8_t fb0__ = rn_tr_frag(q_[4], q_[5]), fb1__ = rn_tr_frag(q_[6], q_[7]);             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    acc[0][0] = RN_MFMA_32x32x16(fa0__, fb0__, acc[0][0], 0, 0, 0);                                \
    acc[0][1] = RN_MFMA_32x32x16(fa0__, fb1__, acc[0][1], 0, 0, 0);                                \
    acc[1][0] = RN_MFMA_32x32x16(fa1__, fb0__, acc[1][0], 0, 0, 0);                                \
    acc[1][1] = RN_MFMA_32x32x16(fa1__, fb1__, acc[1][1], 0, 0, 0);                                \
  } while (0)

  // K loop.  Two LDS rings with counted waits: dy in TWO stages, x in THREE (80 KB: two workgroups per CU).  Per step: dy of
  // step kt + 1, THEN x of step kt + 2 are issued, so that "everything but the last four pieces" (loads retire in order) is
  // exactly what step kt needs; one barrier per step.  (Until round 5: two whole stages and `s_waitcnt vmcnt(0)` before every
  // barrier — each step waited for the round trip of the tile issued at its own top, as the 128-row forward kernel did.)
  WG_ISSUE_DY(0, p_begin);
  WG_ISSUE_X(0, p_begin);
  if (1 < ksteps) WG_ISSUE_X(1, p_begin + WG_BK);
  int cur_x = 0;
#pragma unroll 1
  for (int kt = 0; kt < ksteps; ++kt) {
    if (kt + 1 < ksteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the x pieces of step kt + 1
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everyone's pieces of step kt have landed; everyone is done reading step kt - 1
    asm volatile("" ::: "memory");
    if (kt + 1 < ksteps) WG_ISSUE_DY((kt + 1) & 1, p_begin + (kt + 1) * WG_BK);
    if (kt + 2 < ksteps) WG_ISSUE_X(cur_x >= 1 ? cur_x - 1 : 2, p_begin + (kt + 2) * WG_BK);   // (cur_x + 2) % 3
    const unsigned st_dy = lds0 + (unsigned)((kt & 1) * WG_TILE_BYTES);
    const unsigned st_x = lds0 + (unsigned)((2 + cur_x) * WG_TILE_BYTES);
    // four K slices, reads of slice k + 1 in flight under the MFMAs of slice k (LDS reads return in order: a counted wait)
    rn_u32x2_t qa[8], qb[8];
    WG_READS(qa, st, 0);
    WG_READS(qb, st, 1);
    WG_WAIT(qa, 8);
    WG_MFMAS(qa);
    WG_READS(qa, st, 2);
    WG_WAIT(qb, 8);
    WG_MFMAS(qb);
    WG_READS(qb, st, 3);
    WG_WAIT(qa, 8);
    WG_MFMAS(qa);
    WG_WAIT(qb, 0);
    WG_MFMAS(qb);
    cur_x = cur_x == 2 ? 0 : cur_x + 1;
  }
#undef WG_MFMAS
#undef WG_WAIT
#undef WG_READS
#undef WG_ISSUE_X
#undef WG_ISSUE_DY

  // partial tile -> workspace[chunk][co][tap][ci]
  const int taps = args.R * args.S;
  float* out = args.ws + (long long)chunk * Cout * taps * Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wave_n * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int co = co0 + wave_m * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (co < Cout && ci < Cin) out[((long long)co * taps + tap) * Cin + ci] = acc[i][j][q];
      }
    }
}

// Ordered split-K reduction: dw[i] = sum_c ws[c][i] (+ beta * dw[i]).  256 threads = 64 float4 columns x 4 chunk
// lanes: lane q sums chunks q, q+4, q+8, ... (independent loads, a quarter of the serial chain of the one-thread-per
// -column form, which was latency bound at ~22 us per launch), then the four partial sums are added in lane order —
// a fixed tree, so the result does not depend on the launch geometry.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float4* __restrict__ ws_all, long long n4, int chunks, const WgDwPtrs dws, float beta) {
  __shared__ float4 part[4][64];
  // blockIdx.y = layer of a grouped launch: its partials are [group][chunk][n4], its output dws.p[group]
  const float4* __restrict__ ws = ws_all + (long long)blockIdx.y * chunks * n4;
  float4* __restrict__ dw = dws.p[blockIdx.y];
  const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (long long base = blockIdx.x * 64ll; base < n4; base += (long long)gridDim.x * 64) {
    const long long i = base + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
#pragma unroll 4
      for (int c = q; c < chunks; c += 4) {
        const float4 v = ws[(long long)c * n4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    part[q][col] = a;
    __syncthreads();
    if (q == 0 && i < n4) {
      float4 r = part[0][col];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float4 v = part[k][col];
        r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
      }
      if (beta != 0.0f) {
        const float4 o = dw[i];
        r.x += beta * o.x; r.y += beta * o.y; r.z += beta * o.z; r.w += beta * o.w;
      }
      dw[i] = r;
    }
    __syncthreads();
  }
}

// workgroups a launch aims for (2 per CU x 256 CUs); fewer = fewer split-K partials to write and reduce, more = better
// balance.  Tunable per call (rn_launch_opts.wgrad_target_blocks) for A/B timing.
// (rn_launch_opts.wgrad_target_blocks overrides it, .wgrad_kernel picks the kernel family.)

static int wgrad_plan(const rn_wgrad_problem* p, WgArgs& a) {
  if (!p || p->num_segments < 1 || p->num_segments > RN_CONV_MAX_SEGMENTS) return -1;
  if (p->R < 1 || p->S < 1 || p->stride_h < 1 || p->stride_w < 1) return -1;
  if (rn_validate_launch_opts(p->opts, "rn_conv2d_nhwc_wgrad")) return -1;
  a.R = p->R; a.S = p->S; a.sh = p->stride_h; a.sw = p->stride_w; a.pt = p->pad_top; a.pl = p->pad_left;
  a.nseg = p->num_segments;
  a.Cin = p->seg[0].Cin; a.Cout = p->seg[0].Cout;
  if (a.Cin % 8 || a.Cout % 4 || a.Cin <= 0 || a.Cout <= 0) return -1;
  a.co_tiles = (int)rn_cdiv(a.Cout, 128);
  a.ci_tiles = (int)rn_cdiv(a.Cin, 128);
  long long Ptot = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_wgrad_segment& s = p->seg[i];
    if (!s.x || !s.dy || s.Cin != a.Cin || s.Cout != a.Cout) return -1;
    const long long P = (long long)s.N * s.Ho * s.Wo;
    if (P <= 0 || P >= (1ll << 24)) return -1;
    const long long dyS = s.dy_pix_stride > 0 ? s.dy_pix_stride : s.Cout;
    if (dyS < s.Cout || (dyS % 4)) return -1;
    const long long xS = s.x_pix_stride > 0 ? s.x_pix_stride : s.Cin;
    if (xS % 4) return -1;
    if ((long long)s.N * s.H * s.W * xS * 2 >= (1ll << 31) || P * dyS * 2 >= (1ll << 31)) return -1;
    if ((long long)s.N * s.H * s.W >= (1ll << 24) || xS >= (1 << 24) || dyS >= (1 << 24)) return -1;   // 24-bit multiplies
    Ptot += P;
  }
  const int tiles = a.co_tiles * a.ci_tiles * a.R * a.S;
  a.co_groups = (int)rn_cdiv(tiles, 64);
  if (a.co_groups > a.co_tiles) a.co_groups = a.co_tiles;
  a.gco = (int)rn_cdiv(a.co_tiles, a.co_groups);
  a.co_groups = (int)rn_cdiv(a.co_tiles, a.gco);
  // 512 = one round of two workgroups per CU: measured (tools/bench_wgrad.py, same process) 58 vs 75 us on the 128 <-> 512
  // 1x1 layers of ResNet stage 2 and 41 vs 55 us on 2048 -> 512 against the former 1024 — these layers are HBM-bound and
  // every extra pixel chunk is another |W| x 4 bytes of partial tile written and read back
  long long target = rn_cdiv(p->opts.wgrad_target_blocks > 0 ? p->opts.wgrad_target_blocks : 512, tiles);
  if (target < 1) target = 1;
  if (target > 256) target = 256;
  long long CH = rn_cdiv(rn_cdiv(Ptot, target), WG_BK) * WG_BK;
  if (CH < WG_BK) CH = WG_BK;
  a.CH = (int)CH;
  int chunks = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_wgrad_segment& s = p->seg[i];
    WgSegDev& d = a.seg[i];
    d.x = (const uint16_t*)s.x; d.dy = (const uint16_t*)s.dy;
    d.N = s.N; d.H = s.H; d.W = s.W; d.Ho = s.Ho; d.Wo = s.Wo;
    d.P = s.N * s.Ho * s.Wo;
    d.chunk_begin = chunks;
    d.dyS = s.dy_pix_stride > 0 ? s.dy_pix_stride : s.Cout;
    d.xS = s.x_pix_stride > 0 ? s.x_pix_stride : s.Cin;
    d.pad_ = 0;
    chunks += (int)rn_cdiv(d.P, CH);
  }
  a.total_chunks = chunks;
  a.pad_ = 0;
  // large layers: 256 x 256 per-tap tiles on the ping-pong kernel (rn_wgrad_big.hip); pad_ = 1 marks the choice
  if (p->opts.wgrad_kernel != 1 && rn_wgrad_big_plan(p, a)) a.pad_ = 1;
  return 0;
}

/* which kernel rn_conv2d_nhwc_wgrad runs for `problem`: 0 = wgrad_kernel (128 x 128 per-tap tiles), 1 = wgrad_big_kernel
 * (256 x 256 per-tap tiles), 2 = wgrad_halo_kernel (3x3 / stride 1: all nine taps per workgroup); -1 on a malformed
 * problem.  Profiling / bench bookkeeping only. */
extern "C" int rn_wgrad_kernel_id(const rn_wgrad_problem* p) {
  WgArgs a;
  if (wgrad_plan(p, a)) return -1;
  WhArgs h;
  if (rn_wgrad_halo_plan(&p, 1, h)) return 2;
  return a.pad_;
}

extern "C" size_t rn_wgrad_workspace_bytes(const rn_wgrad_problem* p) {
  WgArgs a;
  if (wgrad_plan(p, a)) return 0;
  WhArgs h;
  if (rn_wgrad_halo_plan(&p, 1, h)) return rn_wgrad_halo_workspace_bytes(h);
  return (size_t)a.total_chunks * a.Cout * a.R * a.S * a.Cin * sizeof(float);
}

// partial-tile launch of a planned problem (wgrad_big_kernel or wgrad_kernel) + the ordered reduction.  ngroups > 1: the
// problem's segments are LAYERS of a grouped call (one segment each, equal chunk counts): segment g's partial tiles are the
// chunks [g * total_chunks / ngroups, ...) of the workspace and its sum goes to dws.p[g].
static int wgrad_launch_planned(WgArgs& a, const rn_launch_opts& opts, void* workspace, hipStream_t st, const WgDwPtrs& dws,
                                int ngroups, float beta) {
  a.ws = (float*)workspace;
  const long long n4 = (long long)a.Cout * a.R * a.S * a.Cin / 4;
  int blocks = (int)(rn_cdiv(n4, 64) < 4096 ? rn_cdiv(n4, 64) : 4096);
  if (blocks * ngroups > 8192) blocks = 8192 / ngroups;
  if (a.pad_ == 1) {
    const int rc = rn_launch_wgrad_big(a, opts, st);
    if (rc != RN_OK) return rc;
  } else {
    const int lds = 5 * WG_TILE_BYTES;
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid((unsigned)(a.gco * a.ci_tiles * a.R * a.S * a.co_groups * a.total_chunks));
    bool linear = a.sh == 1 && a.sw == 1 && a.pt == (a.R - 1) / 2 && a.pl == (a.S - 1) / 2 && (a.R & 1) && (a.S & 1);
    for (int i = 0; i < a.nseg; ++i) {
      const WgSegDev& s = a.seg[i];
      linear = linear && s.Ho == s.H && s.Wo == s.W &&
               (long long)s.N * s.H * s.W * (s.xS > s.dyS ? s.xS : s.dyS) * 2 < (1ll << 31) - (1ll << 24);
    }
    if (linear) hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(WG_THREADS), lds, st, a);
    else hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(WG_THREADS), lds, st, a);
    RN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks, ngroups), dim3(256), 0, st, (const float4*)workspace, n4,
                     a.total_chunks / ngroups, dws, beta);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

extern "C" int rn_conv2d_nhwc_wgrad(const rn_wgrad_problem* p, float* dw, float beta, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  WgArgs a;
  RN_CHECK_ARG(wgrad_plan(p, a) == 0, "rn_conv2d_nhwc_wgrad: bad problem (Cin %% 8, Cout %% 4, < 2^24 pixels, < 2 GiB tensors)");
  RN_CHECK_ARG(dw != nullptr, "rn_conv2d_nhwc_wgrad: null dw");
  const size_t need = rn_wgrad_workspace_bytes(p);
  if (!workspace || workspace_bytes < need) {
    rn_set_error("rn_conv2d_nhwc_wgrad: workspace %zu < %zu", workspace_bytes, need);
    return RN_ENOMEM;
  }
  {
    WhArgs h;
    if (rn_wgrad_halo_plan(&p, 1, h)) {
      h.ws = (float*)workspace;
      const int rc = rn_launch_wgrad_halo(h, p->opts, (hipStream_t)stream);
      if (rc != RN_OK) return rc;
      const long long nb4 = (long long)h.Cout * 9 * h.Cin / 4;
      int blocksh = (int)(rn_cdiv(nb4, 64) < 4096 ? rn_cdiv(nb4, 64) : 4096);
      WgDwPtrs dws;
      dws.p[0] = (float4*)dw;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocksh), dim3(256), 0, (hipStream_t)stream, (const float4*)workspace, nb4,
                         h.total_chunks, dws, beta);
      RN_CHECK_LAUNCH();
      return RN_OK;
    }
  }
  WgDwPtrs one;
  one.p[0] = (float4*)dw;
  return wgrad_launch_planned(a, p->opts, workspace, (hipStream_t)stream, one, 1, beta);
}

// ---- several layers of identical geometry in one launch -----------------------------------------------------------
// Equivalent to n calls of rn_conv2d_nhwc_wgrad.  When the layers share their geometry and wgrad_halo_kernel serves it
// (the eight head-tower layers, the 3x3 layers of a ResNet stage), they run as ONE launch whose tiles are
// (layer, co tile, ci tile): the split-K plan then cuts the pixels into 1/n of the chunks per layer — every workgroup
// writes its whole accumulator (288 KB) as a partial tile, so a launch costs 75 MB of partials however small the layer —
// and one reduction launch sums all layers.  Anything else falls back to the per-layer calls.
// Round 6: layers that wgrad_halo_kernel does not serve (the 1x1 layers of a ResNet stage: five / six of identical geometry)
// group too, WITHOUT another kernel: a layer of one segment becomes a SEGMENT of a merged problem.  The partial-tile kernels
// already walk per-segment (x, dy) pointers and chunk ranges, and with identical geometry every segment gets the same
// number of chunks, so segment g's partial tiles are a contiguous [chunks / n] slice of the workspace — exactly the
// [group][chunk] layout the grouped reduction reads.  Same split-K effect as the halo group: the plan aims for the same
// number of workgroups over n layers' pixels, i.e. 1/n of the partial tiles per layer, one launch + one reduction for n.
static bool wgrad_layers_as_segments(const rn_wgrad_problem* const* ps, int n, rn_wgrad_problem& m) {
  if (!ps || n < 2 || n > RN_WGRAD_MAX_GROUP || n > RN_CONV_MAX_SEGMENTS) return false;
  const rn_wgrad_problem& p0 = *ps[0];
  if (p0.num_segments != 1) return false;
  m = p0;
  m.num_segments = n;
  const rn_wgrad_segment& s0 = p0.seg[0];
  for (int i = 0; i < n; ++i) {
    const rn_wgrad_problem& q = *ps[i];
    if (q.num_segments != 1 || q.R != p0.R || q.S != p0.S || q.stride_h != p0.stride_h || q.stride_w != p0.stride_w ||
        q.pad_top != p0.pad_top || q.pad_left != p0.pad_left || memcmp(&q.opts, &p0.opts, sizeof(rn_launch_opts)) != 0)
      return false;
    const rn_wgrad_segment& s = q.seg[0];
    if (s.N != s0.N || s.H != s0.H || s.W != s0.W || s.Cin != s0.Cin || s.Ho != s0.Ho || s.Wo != s0.Wo || s.Cout != s0.Cout ||
        s.dy_pix_stride != s0.dy_pix_stride || s.x_pix_stride != s0.x_pix_stride)
      return false;
    m.seg[i] = s;
  }
  WgArgs a;
  if (wgrad_plan(&m, a) || a.total_chunks % n) return false;
  for (int i = 0; i < n; ++i)
    if (a.seg[i].chunk_begin != i * (a.total_chunks / n)) return false;
  return true;
}

extern "C" size_t rn_wgrad_group_workspace_bytes(const rn_wgrad_problem* const* ps, int n) {
  if (!ps || n < 1) return 0;
  size_t need = 0;
  for (int i = 0; i < n; ++i) {
    if (!ps[i]) return 0;
    const size_t b = rn_wgrad_workspace_bytes(ps[i]);
    if (b == 0) return 0;
    need = b > need ? b : need;
  }
  WhArgs h;
  if (n > 1 && n <= RN_WGRAD_MAX_GROUP && rn_wgrad_halo_plan(ps, n, h)) {
    const size_t g = rn_wgrad_halo_workspace_bytes(h);
    need = g > need ? g : need;
  } else {
    rn_wgrad_problem m;
    if (wgrad_layers_as_segments(ps, n, m)) {
      const size_t g = rn_wgrad_workspace_bytes(&m);
      need = g > need ? g : need;
    }
  }
  return need;
}

extern "C" int rn_wgrad_group_fused(const rn_wgrad_problem* const* ps, int n) {   // 1: one grouped launch, 0: per-layer calls
  if (!ps || n < 2 || n > RN_WGRAD_MAX_GROUP) return 0;
  for (int i = 0; i < n; ++i)
    if (!ps[i]) return 0;
  WhArgs h;
  if (rn_wgrad_halo_plan(ps, n, h)) return 1;
  rn_wgrad_problem m;
  return wgrad_layers_as_segments(ps, n, m) ? 1 : 0;
}

extern "C" int rn_conv2d_nhwc_wgrad_group(const rn_wgrad_problem* const* ps, int n, float* const* dws, float beta,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  RN_CHECK_ARG(ps && dws && n >= 1, "rn_conv2d_nhwc_wgrad_group: bad argument");
  for (int i = 0; i < n; ++i) RN_CHECK_ARG(ps[i] && dws[i], "rn_conv2d_nhwc_wgrad_group: null problem / output %d", i);
  const size_t need = rn_wgrad_group_workspace_bytes(ps, n);
  RN_CHECK_ARG(need > 0, "rn_conv2d_nhwc_wgrad_group: bad problem");
  if (!workspace || workspace_bytes < need) {
    rn_set_error("rn_conv2d_nhwc_wgrad_group: workspace %zu < %zu", workspace_bytes, need);
    return RN_ENOMEM;
  }
  WhArgs h;
  if (n > 1 && n <= RN_WGRAD_MAX_GROUP && rn_wgrad_halo_plan(ps, n, h)) {
    h.ws = (float*)workspace;
    const int rc = rn_launch_wgrad_halo(h, ps[0]->opts, (hipStream_t)stream);
    if (rc != RN_OK) return rc;
    const long long nb4 = (long long)h.Cout * 9 * h.Cin / 4;
    int blocks = (int)(rn_cdiv(nb4, 64) < 4096 ? rn_cdiv(nb4, 64) : 4096);
    if (blocks * n > 8192) blocks = 8192 / n;
    WgDwPtrs d;
    for (int i = 0; i < n; ++i) d.p[i] = (float4*)dws[i];
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, (const float4*)workspace, nb4,
                       h.total_chunks, d, beta);
    RN_CHECK_LAUNCH();
    return RN_OK;
  }
  rn_wgrad_problem m;
  if (wgrad_layers_as_segments(ps, n, m)) {
    WgArgs a;
    RN_CHECK_ARG(wgrad_plan(&m, a) == 0, "rn_conv2d_nhwc_wgrad_group: bad merged problem");
    WgDwPtrs d;
    for (int i = 0; i < n; ++i) d.p[i] = (float4*)dws[i];
    return wgrad_launch_planned(a, m.opts, workspace, (hipStream_t)stream, d, n, beta);
  }
  for (int i = 0; i < n; ++i) {
    const int rc = rn_conv2d_nhwc_wgrad(ps[i], dws[i], beta, workspace, workspace_bytes, stream);
    if (rc != RN_OK) return rc;
  }
  return RN_OK;
}
