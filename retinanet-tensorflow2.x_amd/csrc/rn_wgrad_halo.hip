// rn_wgrad_halo.hip — weight gradient of the 3x3 / stride 1 / pad 1 layers (head towers, class prediction, FPN output
// convs, the 3x3 of every stride-1 bottleneck: ~70 % of the weight-gradient FLOPs of the step) with ALL NINE TAPS in one
// workgroup.
//
//   dW[co][r][s][ci] = sum over output pixels (n, y, x) of dy[n][y][x][co] * x_in[n][y + r - 1][x + s - 1][ci]
//
// The per-tap kernels (rn_wgrad.hip, rn_wgrad_big.hip) re-stage a dy tile and a shifted x tile for every tap: 32 KB of
// LDS-DMA and 24 transpose reads per 16 MFMAs per wave — the load segment, not the matrix pipe, sets their pace
// (MFMA busy 0.36, profiles/traffic.json).  Here the taps share what they can share:
//   * the reduction runs over IMAGE ROWS: one K slice = 16 consecutive pixels of one row (a column tile of 16), and the
//     stream walks down the rows of a column strip.  With G the padded row index (G = n*(H+1) + y + 1: ONE shared zero
//     row between images, as in rn_conv_halo.hip) every row does
//         acc[r][s] += D[G + 1 - r]^T * X[G][s]      r, s = 0..2
//     where X[G][s] is the fragment of input row G shifted by s - 1 columns and D[.] the fragment of a dy row: pad
//     rows are zero rows, so image borders need no masks and no branches;
//   * an input row is staged ONCE (18 pixels: the tile's 16 + one halo column each side; columns outside the image
//     are zero-filled by the DMA) and read as three shifted fragments; a dy row is staged once, read once and then
//     lives in registers for three rows (the r = 0, 1, 2 products of rows G-1, G, G+1 rotate through Dp1 -> D0 -> Dm1);
//   * wave tile = 32 (co) x 32 (ci) x 9 taps = 9 MFMA tiles (144 accumulator registers); workgroup = 8 waves as
//     4 (co) x 2 (ci) = 128 x 64 x 9; per step of TWO rows a wave issues 18 MFMAs against 16 transpose reads
//     (0.89 per MFMA instead of 1.5) and the workgroup stages 13 KB (8 + 5 DMA pieces instead of 32 per 128 MFMAs);
//   * the two waves of a SIMD ping-pong exactly as in wgrad_big_kernel / conv_big_kernel: compute segment = 18 MFMAs
//     from registers, load segment = fragment reads of the next step + this wave's DMA pieces three steps ahead +
//     counted vmcnt; two s_barriers per step; eight LDS stages of 13 KB (seven steps in flight);
//   * the DMA source offset of a lane is a per-strip constant (its column / channel slot) plus a per-row scalar: ~5
//     VALU per piece (the per-tap kernels spend 17-45 VALU per DMA row pair on (n, y, x) bookkeeping).
// LDS images (lane-linear DMA destinations, swizzle on the SOURCE channel slot and on the read):
//   dy rows  [16 px][128 co]  256 B per pixel, 16-byte slot ^ ((px & 3) << 2)     (4 pixels of a transpose read -> 4 bank groups)
//   x rows   [18 px][ 64 ci]  128 B per pixel, 16-byte slot ^ (((pc >> 1) & 1) << 2) (any 4 consecutive pixels -> 4 bank groups)
// Work decomposition: the rows of all (segment, column strip)s form one step sequence (every strip opens with a
// load-only step that brings its first two dy rows); a workgroup = (co tile, ci tile, chunk of consecutive steps) and
// writes its 128 x 9 x 64 partial tile into the same workspace layout as the per-tap kernels
// ([chunk][co][tap][ci]), summed in index order by wgrad_reduce_kernel: deterministic, no float atomics.
#include <string.h>

#include <algorithm>

#include "rn_wgrad_dev.h"

namespace {

constexpr int WH_STAGES_DEFAULT = 8;   // LDS ring: steps in flight = stages - 1 (the data streams from HBM: latency, not rate)
constexpr int WH_DY_BYTES = 2 * 16 * 256;          // two dy rows of 16 pixels x 128 channels
constexpr int WH_X_BYTES = 5 * 1024;               // two x rows of 18 pixels x 64 channels (4608 B) in five DMA pieces
constexpr int WH_STAGE = WH_DY_BYTES + WH_X_BYTES;  // 13 KB

__device__ __forceinline__ void wh_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

// VAR (probe builds only, -DRN_PROBES: rn_launch_opts.ablate picks it, tools/bench_wgrad.py times the variants in one
// process): bit 0 = transpose reads through the builtin (the compiler then drains the DMA ring in front of them), bit 1 =
// no DMA issue, bit 2 = no fragment reads, bit 3 = no MFMAs — bits 1-3 give wrong results, timing only.
template <int VAR, int WH_STAGES = WH_STAGES_DEFAULT>
__global__ void __launch_bounds__(512) wgrad_halo_kernel(const WhArgs args) {
  static_assert((WH_STAGES & (WH_STAGES - 1)) == 0 && WH_STAGES >= 4, "ring stages: a power of two");
  constexpr int PD = WH_STAGES - 1;    // prefetch distance in steps
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;       // 4 (co) x 2 (ci) waves; wn is also the ping-pong group
  const int tiles_g = args.co_tiles * args.ci_tiles;     // tiles of one layer
  const int tiles = tiles_g * args.ngroups;
  const int items = tiles * args.total_chunks;
  const int Cin = args.Cin, Cout = args.Cout;
  const unsigned lds0 = rn_lds_addr(smem);

  // fragment read offsets inside a stage (constant for the whole kernel): lane (g, i) of a 16-lane group reads the
  // 4-pixel x 4-channel block at pixel (g>>1)*8 + (i>>2) [+4], channels 16*(g&1) + 4*(i&3) of its wave's 32-channel block
  int offd, offx[3];
  {
    const int g = lane >> 4, i = lane & 15;
    const int px = (g >> 1) * 8 + (i >> 2);
    const int cd = wm * 32 + 16 * (g & 1) + 4 * (i & 3);
    offd = px * 256 + ((((cd >> 3) ^ ((px & 3) << 2))) << 4) + (cd & 7) * 2;
    const int cx = wn * 32 + 16 * (g & 1) + 4 * (i & 3);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int pc = px + s;
      offx[s] = WH_DY_BYTES + pc * 128 + ((((cx >> 3) ^ (((pc >> 1) & 1) << 2))) << 4) + (cx & 7) * 2;
    }
  }
  // DMA geometry of this lane.  dy: piece `wave` = row (wave >> 2), pixels 4*(wave & 3) .. +3, lane = (px & 3)*16 + slot.
  // x: pieces 0..4 by waves 0..4, piece q = patch pixels 8q .. 8q+7 (patch pixel = row*18 + column), lane = px*8 + slot.
  const int dy_px = 4 * (wave & 3) + (lane >> 4);
  const int dy_slot = (lane & 15) ^ ((dy_px & 3) << 2);          // source channel slot (8 channels each)
  const int x_pp = 8 * wave + (lane >> 3);                        // 0..39; >= 36: padding of the last piece
  const int x_row = x_pp >= 18 ? 1 : 0, x_pc = x_pp - 18 * x_row;
  const int x_slot = (lane & 7) ^ (((x_pc >> 1) & 1) << 2);

#pragma unroll 1
  for (int bid = blockIdx.x; bid < items; bid += gridDim.x) {
    int logical;
    {   // XCD-aware: the tiles of one chunk are consecutive on one XCD (they walk the same rows: L2 reuse)
      const int xcd = bid & 7, slot = bid >> 3;
      const int q = items >> 3, rr = items & 7;
      logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
    }
    const int chunk = logical / tiles;
    const int tile = logical - chunk * tiles;
    const int grp = tile / tiles_g, tile_g = tile - grp * tiles_g;
    const int co_t = tile_g / args.ci_tiles, ci_t = tile_g - co_t * args.ci_tiles;
    const int co0 = co_t * 128, ci0 = ci_t * 64;
    const int step_a = chunk * args.CHs;
    const int step_b = step_a + args.CHs < args.total_steps ? step_a + args.CHs : args.total_steps;
    const int first = step_a > 0 ? step_a - 1 : 0;       // one leading load-only step brings the dy rows the chunk starts with
    const int nsteps = step_b - first;

    // ---- issue side: the step whose rows are being DMA'd -------------------------------------------------------
    int i_si = 0, i_ct = 0, i_t = 0;          // segment, column tile, step inside the strip (0 = the strip's load-only step)
#pragma unroll 1
    for (int k = 1; k < args.nseg; ++k)
      if (first >= args.seg[k].step_begin) i_si = k;
    {
      const int rem = first - args.seg[i_si].step_begin;
      i_ct = rem / args.seg[i_si].L;
      i_t = rem - i_ct * args.seg[i_si].L;
    }
    int i_N = 0, i_H1 = 0, i_L = 0, i_ctiles = 0;
    unsigned i_xrow = 0, i_dyrow = 0;         // bytes per image row of x / dy
    unsigned l_dy = WG_OOB, l_x = WG_OOB;     // this lane's column / channel-slot offset inside a row, or out of range
    // Row trackers.  A step with first padded row G0 = 2t stages the x rows G0-1, G0 and the dy rows G0, G0+1; this wave
    // issues ONE dy row (G0 + wave/4) and lanes of the two x rows.  A tracked row G = n*(H+1) + yy is a real image row iff
    // yy != 0 and 0 <= n < N; its byte offset (G - n - 1) * row bytes advances by two rows per step, by one when the
    // step crosses an image seam — a handful of scalar adds per step (recomputing (n, yy, offset) with multiplies for
    // four rows cost every wave ~60 scalar instructions per step: the load segment, not the MFMAs, set the pace).
    int d_yy = 0, d_n = 0, a_yy = 0, a_n = 0, b_yy = 0, b_n = 0;
    unsigned d_off = 0, a_off = 0, b_off = 0;
    __amdgpu_buffer_rsrc_t rs_dy, rs_x;
#define WH_TRK_INIT(yy_, n_, off_, G_, rowb_)                                     \
  do {                                                                            \
    const int G__ = (G_);                                                         \
    n_ = G__ < 0 ? -1 : G__ / i_H1;                                               \
    yy_ = G__ - n_ * i_H1;                                                        \
    off_ = (unsigned)(G__ - n_ - 1) * (rowb_);                                    \
  } while (0)
#define WH_TRK_ADV(yy_, n_, off_, rowb_)                                          \
  do {                                                                            \
    yy_ += 2;                                                                     \
    const bool w__ = yy_ >= i_H1;                                                 \
    yy_ -= w__ ? i_H1 : 0;                                                        \
    n_ += w__ ? 1 : 0;                                                            \
    off_ += w__ ? (rowb_) : 2 * (rowb_);                                          \
  } while (0)
#define WH_TRK_OFF(yy_, n_, off_) (((yy_) != 0 && (unsigned)(n_) < (unsigned)i_N) ? (off_) : WG_OOB)
#define WH_STRIP_SETUP()                                                                                   \
  do {                                                                                                     \
    const WhSeg& sg__ = args.seg[i_si];                                                                    \
    i_N = sg__.N; i_H1 = sg__.H + 1; i_L = sg__.L; i_ctiles = sg__.ctiles;                                 \
    i_xrow = (unsigned)(sg__.W * sg__.xS * 2); i_dyrow = (unsigned)(sg__.W * sg__.dyS * 2);               \
    const WhPtr& pt__ = args.ptr[grp][i_si];                                                               \
    rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)pt__.dy, 0,                                           \
                                              (int)((long long)sg__.N * sg__.H * sg__.W * sg__.dyS * 2), 0x00020000); \
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)pt__.x, 0,                                             \
                                             (int)((long long)sg__.N * sg__.H * sg__.W * sg__.xS * 2), 0x00020000);  \
    const int x0__ = i_ct * 16;                                                                            \
    const int cdy__ = co0 + dy_slot * 8, cx__ = ci0 + x_slot * 8;                                          \
    l_dy = (x0__ + dy_px < sg__.W && cdy__ < Cout) ? (unsigned)(((x0__ + dy_px) * sg__.dyS + cdy__) * 2) : WG_OOB; \
    const int xc__ = x0__ - 1 + x_pc;                                                                      \
    l_x = (x_pp < 36 && (unsigned)xc__ < (unsigned)sg__.W && cx__ < Cin) ? (unsigned)((xc__ * sg__.xS + cx__) * 2) : WG_OOB; \
    /* step t of a strip covers the padded rows G = 2t - 1, 2t (x) and 2t, 2t + 1 (dy): G0 = 2t */         \
    WH_TRK_INIT(d_yy, d_n, d_off, 2 * i_t + (wave >> 2), i_dyrow);                                         \
    WH_TRK_INIT(a_yy, a_n, a_off, 2 * i_t - 1, i_xrow);                                                    \
    WH_TRK_INIT(b_yy, b_n, b_off, 2 * i_t, i_xrow);                                                        \
  } while (0)
    WH_STRIP_SETUP();

    // Load-only steps need no branch around the MFMAs (a conditional block makes the register allocator copy all 144
    // accumulators at the join): their x rows are ZERO in LDS — a strip's step 0 reads the pad rows -1 and 0 anyway, and
    // the chunk's leading step gets out-of-range x offsets — so their products vanish.
    // The address work of a step is split off the DMA instructions: WH_PREP() turns the trackers into this lane's two
    // source offsets for the NEXT issue and advances the stream; it runs inside the COMPUTE segment, where a wave issues one
    // MFMA per 32 cycles and the scalar / vector bookkeeping rides in the gaps.  The load segment — whose instruction
    // count sets the pace of the ping-pong (a wave issues in order, ~5 cycles per instruction) — keeps only the fragment
    // reads, two DMA instructions and the waits.
    int g_iss = 0, g_prep = 0;
    unsigned v_dy = WG_OOB, v_x = WG_OOB;
/* (the stream state — trackers, lane offsets, the segment's buffer descriptors — describes the PREPARED step until the \
   next WH_PREP: the issue of that step still needs the descriptors, so the advance opens the next call) */ \
#define WH_PREP()                                                                                          \
  do {                                                                                                     \
    if (g_prep != 0) {                                                                                     \
      if (++i_t == i_L) {             /* next column strip (or segment) */                                 \
        i_t = 0;                                                                                           \
        if (++i_ct == i_ctiles) { i_ct = 0; if (i_si + 1 < args.nseg) ++i_si; }                            \
        WH_STRIP_SETUP();                                                                                  \
      } else {                                                                                             \
        WH_TRK_ADV(d_yy, d_n, d_off, i_dyrow);                                                             \
        WH_TRK_ADV(a_yy, a_n, a_off, i_xrow);                                                              \
        WH_TRK_ADV(b_yy, b_n, b_off, i_xrow);                                                              \
      }                                                                                                    \
    }                                                                                                      \
    const unsigned rdy__ = WH_TRK_OFF(d_yy, d_n, d_off);                                                   \
    v_dy = (l_dy == WG_OOB || rdy__ == WG_OOB) ? WG_OOB : l_dy + rdy__;                                    \
    const unsigned xa__ = WH_TRK_OFF(a_yy, a_n, a_off), xb__ = WH_TRK_OFF(b_yy, b_n, b_off);               \
    const unsigned rx__ = g_prep == 0 ? WG_OOB : (x_row ? xb__ : xa__);   /* leading step: dy rows only */  \
    v_x = (l_x == WG_OOB || rx__ == WG_OOB) ? WG_OOB : l_x + rx__;                                         \
    asm volatile("" : "+v"(v_dy), "+v"(v_x));   /* materialise HERE: the compiler would sink the selects to the DMA */ \
    ++g_prep;                                                                                              \
  } while (0)
#define WH_ISSUE()                                                                                         \
  do {                                                                                                     \
    char* st__ = smem + (g_iss & (WH_STAGES - 1)) * WH_STAGE;                                              \
    if (!(VAR & 2)) wh_dma16(rs_dy, st__ + wave * 1024, v_dy);                                             \
    if (wave < 5 && !(VAR & 2)) wh_dma16(rs_x, st__ + WH_DY_BYTES + wave * 1024, v_x);                     \
    ++g_iss;                                                                                               \
  } while (0)

    f32x16_t acc[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][s][q] = 0.0f;

    // fragments: dy rows G-1, G, G+1, G+2 of the step's first row G (Dm1, D0 carried from the step before), x rows a / b
    bf16x8_t Dm1, D0, Dp1, Dp2, Xa[3], Xb[3];
#pragma unroll
    for (int q = 0; q < 8; ++q) { Dm1[q] = (rn_h16)0.0f; D0[q] = (rn_h16)0.0f; }   // (a zero x row times a NaN bit pattern would be NaN)

// 16 transpose reads from inline asm (rn_wgrad_dev.h: the compiler would drain the DMA ring in front of the builtin), the
// dy pair first; they return while this wave issues its DMA pieces; one wait names every destination
#define WH_READ(stage_)                                                                            \
  do {                                                                                             \
    const unsigned t__ = lds0 + (unsigned)((stage_) * WH_STAGE);                                   \
    const unsigned ad__ = t__ + (unsigned)offd;                                                    \
    const unsigned a0__ = t__ + (unsigned)offx[0], a1__ = t__ + (unsigned)offx[1], a2__ = t__ + (unsigned)offx[2]; \
    RN_TR_ISSUE(q[0], ad__, 0); RN_TR_ISSUE(q[1], ad__, 1024);                                     \
    RN_TR_ISSUE(q[2], ad__, 4096); RN_TR_ISSUE(q[3], ad__, 5120);                                  \
    RN_TR_ISSUE(q[4], a0__, 0); RN_TR_ISSUE(q[5], a0__, 512);                                      \
    RN_TR_ISSUE(q[6], a1__, 0); RN_TR_ISSUE(q[7], a1__, 512);                                      \
    RN_TR_ISSUE(q[8], a2__, 0); RN_TR_ISSUE(q[9], a2__, 512);                                      \
    RN_TR_ISSUE(q[10], a0__, 2304); RN_TR_ISSUE(q[11], a0__, 2816);                                \
    RN_TR_ISSUE(q[12], a1__, 2304); RN_TR_ISSUE(q[13], a1__, 2816);                                \
    RN_TR_ISSUE(q[14], a2__, 2304); RN_TR_ISSUE(q[15], a2__, 2816);                                \
  } while (0)
#define WH_READ_BUILTIN(stage_)                                                                    \
  do {                                                                                             \
    const char* t__ = smem + (stage_) * WH_STAGE;                                                  \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                \
      q[2 * k] = __builtin_bit_cast(rn_u32x2_t, rn_ds_read_tr4((const lds_b4_t*)(t__ + offd + 4096 * k)));          \
      q[2 * k + 1] = __builtin_bit_cast(rn_u32x2_t, rn_ds_read_tr4((const lds_b4_t*)(t__ + offd + 4096 * k + 1024))); \
    }                                                                                              \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                  \
      _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                              \
        q[4 + 6 * k + 2 * s] = __builtin_bit_cast(rn_u32x2_t, rn_ds_read_tr4((const lds_b4_t*)(t__ + offx[s] + 2304 * k)));           \
        q[4 + 6 * k + 2 * s + 1] = __builtin_bit_cast(rn_u32x2_t, rn_ds_read_tr4((const lds_b4_t*)(t__ + offx[s] + 2304 * k + 512))); \
      }                                                                                            \
  } while (0)
#define WH_LOADSEG(stage_)                                                                         \
  do {                                                                                             \
    rn_u32x2_t q[16];                                                                              \
    if (VAR & 4) { _Pragma("unroll") for (int k = 0; k < 16; ++k) q[k] = rn_u32x2_t{0x3c003c00u + (unsigned)k, 0x3c003c00u}; } \
    else if (VAR & 1) WH_READ_BUILTIN(stage_);                                                     \
    else WH_READ(stage_);                                                                          \
    if (g_iss < nsteps) {                                                                          \
      WH_ISSUE();                                                                                  \
      /* everything but the pieces of the last PD - 1 steps (2 or 1 per step and wave) has landed */ \
      if (wave < 5) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * (PD - 1)) : "memory");           \
      else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PD - 1) : "memory");                          \
    } else {                                                                                       \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    }                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                            \
                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]),      \
                   "+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]) \
                 :: "memory");                                                                     \
    Dp1 = rn_tr_frag(q[0], q[1]); Dp2 = rn_tr_frag(q[2], q[3]);                                    \
    Xa[0] = rn_tr_frag(q[4], q[5]); Xa[1] = rn_tr_frag(q[6], q[7]); Xa[2] = rn_tr_frag(q[8], q[9]); \
    Xb[0] = rn_tr_frag(q[10], q[11]); Xb[1] = rn_tr_frag(q[12], q[13]); Xb[2] = rn_tr_frag(q[14], q[15]); \
  } while (0)
// row a = G: r = 0 takes dy row G+1 (Dp1), r = 1 row G (D0), r = 2 row G-1 (Dm1); row b = G+1: Dp2, Dp1, D0
#define WH_COMPUTESEG()                                                                            \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (VAR & 8) { WH_PREP(); break; }                                                             \
    if (VAR & 16) __builtin_amdgcn_s_setprio(1);                                                   \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) acc[0][s] = RN_MFMA_32x32x16(Dp1, Xa[s], acc[0][s], 0, 0, 0); \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) acc[1][s] = RN_MFMA_32x32x16(D0, Xa[s], acc[1][s], 0, 0, 0);  \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) acc[2][s] = RN_MFMA_32x32x16(Dm1, Xa[s], acc[2][s], 0, 0, 0); \
    WH_PREP();                    /* the next issue's addresses, in the MFMA gaps */                 \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) acc[0][s] = RN_MFMA_32x32x16(Dp2, Xb[s], acc[0][s], 0, 0, 0); \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) acc[1][s] = RN_MFMA_32x32x16(Dp1, Xb[s], acc[1][s], 0, 0, 0); \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) acc[2][s] = RN_MFMA_32x32x16(D0, Xb[s], acc[2][s], 0, 0, 0);  \
    Dm1 = Dp1; D0 = Dp2;          /* carries of the next step: rows G+1, G+2 become G-1, G */        \
    if (VAR & 16) __builtin_amdgcn_s_setprio(0);                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#define WH_BARRIER()                        \
  do {                                      \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

    // prologue: steps 0..2 in flight, 0 and 1 complete; pre-roll: both groups read step 0 and issue step 3 (group 1 as
    // its slot-0 load segment).  Both groups execute 2 * nsteps barriers.
#pragma unroll 1
    for (int t = 0; t < PD && g_iss < nsteps; ++t) { WH_PREP(); WH_ISSUE(); }
    WH_PREP();               // the first in-loop issue
    if (g_iss >= PD) {      // steps 0 and 1 complete
      if (wave < 5) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * (PD - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PD - 2) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WH_BARRIER();
    int g = 0;
    WH_LOADSEG(0);
    if (wn == 1) WH_BARRIER();
#pragma unroll 1
    while (true) {
      WH_COMPUTESEG();
      WH_BARRIER();
      if (++g == nsteps) break;
      WH_LOADSEG(g & (WH_STAGES - 1));
      WH_BARRIER();
    }
    if (wn == 0) WH_BARRIER();
#undef WH_BARRIER
#undef WH_COMPUTESEG
#undef WH_LOADSEG
#undef WH_READ
#undef WH_READ_BUILTIN
#undef WH_ISSUE
#undef WH_PREP
#undef WH_TRK_OFF
#undef WH_TRK_ADV
#undef WH_TRK_INIT
#undef WH_STRIP_SETUP

    // partial tile -> workspace[chunk][co][tap][ci] (lane = ci column: 128-byte rows); a fresh lane id so that nothing
    // of the epilogue is live across the loop
    const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int row_stride = 9 * Cin;                      // floats between consecutive co rows
    float* out = args.ws + ((long long)grp * args.total_chunks + chunk) * Cout * row_stride;
    const int ci = ci0 + wn * 32 + (elane & 31);
    const int cob = co0 + wm * 32 + 4 * (elane >> 5);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float* po = out + (long long)cob * row_stride + (r * 3 + s) * Cin + ci;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int dco = (q & 3) + 8 * (q >> 2);
          if (cob + dco < Cout && ci < Cin) po[(long long)dco * row_stride] = acc[r][s][q];
        }
      }
    __syncthreads();   // the next item's DMA reuses the stages
  }   // work items of this workgroup
}

}  // namespace

// Layers this kernel serves: 3x3 / stride 1 / pad 1, same-size output, Cin a multiple of 64, Cout a multiple of 8 and
// at least 64.  Auto-selected when the launch holds at least 16 384 output pixels (rn_launch_opts.wgrad_kernel = 2:
// whatever the pixel count; = 3: keep the per-tap wgrad_big_kernel for A/B timing).
// ps[0..ngroups): layers of IDENTICAL geometry (rn_conv2d_nhwc_wgrad_group) — one launch, tiles = groups x co x ci, so
// the split-K plan needs 1/ngroups of the pixel chunks per layer and writes 1/ngroups of the partial-tile bytes:
// every workgroup of a launch writes its 288 KB accumulator once, 75 MB per launch however small the layer (measured:
// eight head-tower layers as one launch 2.12 ms against 8 x 0.346 ms, tools/bench_wgrad.py presets tower8 / tower).
bool rn_wgrad_halo_plan(const rn_wgrad_problem* const* ps, int ngroups, WhArgs& a) {
  if (ngroups < 1 || ngroups > RN_WGRAD_MAX_GROUP) return false;
  const rn_wgrad_problem* p = ps[0];
  if (p->R != 3 || p->S != 3 || p->stride_h != 1 || p->stride_w != 1 || p->pad_top != 1 || p->pad_left != 1) return false;
  if (p->opts.wgrad_kernel == 1 || p->opts.wgrad_kernel == 3) return false;
  const int Cin = p->seg[0].Cin, Cout = p->seg[0].Cout;
  if (Cin % 64 != 0 || Cout % 8 != 0 || Cout < 64) return false;
  long long Ptot = 0, steps = 0;
  a.nseg = p->num_segments;
  a.ngroups = ngroups;
  a.pad_ = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    const rn_wgrad_segment& s = p->seg[i];
    if (s.Ho != s.H || s.Wo != s.W || !s.x || !s.dy) return false;
    WhSeg& d = a.seg[i];
    d.N = s.N; d.H = s.H; d.W = s.W;
    d.dyS = s.dy_pix_stride > 0 ? s.dy_pix_stride : s.Cout;
    d.xS = s.x_pix_stride > 0 ? s.x_pix_stride : s.Cin;
    d.ctiles = (int)rn_cdiv(s.W, 16);
    // padded rows G = 1 .. N*(H+1) - 1 carry products; step t >= 1 of a strip covers G = 2t - 1, 2t; step 0 is load-only
    d.L = (int)rn_cdiv((long long)s.N * (s.H + 1) - 1, 2) + 1;
    d.step_begin = (int)steps;
    d.pad_ = 0;
    steps += (long long)d.ctiles * d.L;
    Ptot += (long long)s.N * s.H * s.W;
    if ((long long)s.N * s.H * s.W * (d.xS > d.dyS ? d.xS : d.dyS) * 2 >= (1ll << 31) - (1ll << 24)) return false;
    for (int g = 0; g < ngroups; ++g) {      // the other layers: the same geometry, their own tensors
      const rn_wgrad_problem* q = ps[g];
      const rn_wgrad_segment& t = q->seg[i];
      if (q->R != p->R || q->S != p->S || q->stride_h != p->stride_h || q->stride_w != p->stride_w ||
          q->pad_top != p->pad_top || q->pad_left != p->pad_left || q->num_segments != p->num_segments ||
          memcmp(&q->opts, &p->opts, sizeof(p->opts)) != 0 || t.N != s.N || t.H != s.H || t.W != s.W || t.Cin != s.Cin ||
          t.Ho != s.Ho || t.Wo != s.Wo || t.Cout != s.Cout || t.dy_pix_stride != s.dy_pix_stride ||
          t.x_pix_stride != s.x_pix_stride || !t.x || !t.dy)
        return false;
      a.ptr[g][i].x = (const uint16_t*)t.x;
      a.ptr[g][i].dy = (const uint16_t*)t.dy;
    }
  }
  if (Ptot < 16384 && p->opts.wgrad_kernel != 2) return false;
  if (steps >= (1ll << 30)) return false;
  a.Cin = Cin; a.Cout = Cout;
  a.co_tiles = (int)rn_cdiv(Cout, 128);
  a.ci_tiles = Cin / 64;
  a.total_steps = (int)steps;
  const int tiles = a.co_tiles * a.ci_tiles * ngroups;
  // one round of the CUs the kernel may use: fewest split-K partials; a chunk is at least 24 steps long — every chunk
  // writes a 288 KB partial tile — unless the caller asks for MORE workgroups than the chip has (the tests of the chunk
  // seams do: short chunks on purpose).  The two-stream engine's CU cap (wgrad_target_blocks = 176) keeps 24.
  long long blocks = p->opts.wgrad_target_blocks > 0 ? p->opts.wgrad_target_blocks : rn_num_cus() - p->opts.reserved_cus;
  const long long min_steps = p->opts.wgrad_target_blocks > rn_num_cus() ? 2 : 24;
  long long chunks = blocks / tiles;
  if (chunks < 1) chunks = 1;
  if (chunks > rn_cdiv(steps, min_steps)) chunks = rn_cdiv(steps, min_steps);
  a.CHs = (int)rn_cdiv(steps, chunks);
  a.total_chunks = (int)rn_cdiv(steps, a.CHs);
  return true;
}

size_t rn_wgrad_halo_workspace_bytes(const WhArgs& a) {
  return (size_t)a.ngroups * a.total_chunks * a.Cout * 9 * a.Cin * sizeof(float);
}

template <int VAR, int STAGES = WH_STAGES_DEFAULT>
static int wh_launch(const WhArgs& a, dim3 grid, hipStream_t st) {
  static unsigned long long attr_set = 0;   // per template instantiation, one bit per device
  constexpr int WH_LDS = STAGES * WH_STAGE;
  if (RN_ATTRS_NEEDED(attr_set)) {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_halo_kernel<VAR, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, WH_LDS));
    RN_ATTRS_DONE(attr_set);
  }
  hipLaunchKernelGGL((wgrad_halo_kernel<VAR, STAGES>), grid, dim3(512), WH_LDS, st, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

int rn_launch_wgrad_halo(const WhArgs& a, const rn_launch_opts& opts, hipStream_t st) {
  const int items = a.ngroups * a.co_tiles * a.ci_tiles * a.total_chunks;
  rn_launch_opts o = opts;
  o.max_workgroups = 0;   // the cap is for the persistent convolution grids
  dim3 grid((unsigned)(opts.reserved_cus > 0 ? rn_persistent_grid(items, rn_num_cus(), o) : items));
#ifdef RN_PROBES
  switch (opts.ablate) {
    case 1: return wh_launch<1>(a, grid, st);
    case 2: return wh_launch<2>(a, grid, st);
    case 4: return wh_launch<4>(a, grid, st);
    case 6: return wh_launch<6>(a, grid, st);
    case 8: return wh_launch<8>(a, grid, st);
    case 14: return wh_launch<14>(a, grid, st);
    case 16: return wh_launch<16>(a, grid, st);       // s_setprio(1) around the MFMA cluster
    case 32: return wh_launch<0, 4>(a, grid, st);     // ring depth A/B
    default: break;
  }
#endif
  return wh_launch<0>(a, grid, st);
}
