// rn_conv_halo.hip — persistent 256 x 256 x 32 kernel for 3x3 / stride 1 / pad 1 convolutions with
// Cout >= 256 (head towers, class prediction, FPN output convs, ResNet stage 3/4 3x3, and their data gradients).
//
// conv_big_kernel (rn_conv_big.hip) stages 16 KB of pixels + 16 KB of weights per K step, and for a 3x3 filter
// eight of the nine pixel tiles of a channel chunk are the ninth one shifted by a pixel: the LDS-DMA path
// (~45 B/clk/CU, ~100 issue cycles per 1 KB instruction) carries the same bytes nine times and bounds the
// kernel at ~1000 TFLOP/s.  Here the K loop runs chunk-major (32 input channels, then the 9 taps) and the pixels
// of a chunk are staged ONCE, as the halo patch the tile's 256 output pixels need:
//
//   * patch = the input rows from one above the tile's first pixel row to one below its last, with ONE zero
//     column between consecutive rows and ONE zero row between consecutive images (and above the first / below
//     the last): "padded global row" G = n*(H+1) + iy + 1, row stride W+1, patch order
//     p = (G - G0)*(W+1) + ix + 1, 64 B per pixel.  Output pixel m, tap (r,s) reads patch pixel
//     base(m) + r*(W+1) + s: the pixel right of a row's last one IS the zero pixel left of the next row, the
//     row below an image's last one IS the zero row above the next image; borders are ordinary zero pixels
//     (written by the DMA's out-of-range zero fill), there are no per-tap masks, a tap is a uniform offset.
//     Capacity 640 pixels = 40 KB (a 256-pixel tile of an 80 x 80 level that straddles two images needs
//     7 rows x 81 + 1 = 568), two patches (the next chunk's lands while this one is used) + a three-stage ring
//     of 16 KB weight tiles + a 10 KB table of per-thread source offsets = 138 KB.  Staged bytes per K step:
//     16 KB + ~4 KB instead of 32 KB.
//   * per wave and K step: 2 weight pieces, and in steps 1..5 of a chunk one piece of the next chunk's patch
//     (5 x 8 waves x 1 KB = 40 KB), against 4 pieces before.  The nine taps are unrolled, so the counted
//     `s_waitcnt vmcnt` of each load segment is a constant (everything but the last two segments' pieces).
//   * same two-group ping-pong, barriers, XCD-aware persistent tile walk and per-wave epilogue as
//     conv_big_kernel; the epilogue's 32 KB of transpose patches alias the pixel patch of the tile's last chunk,
//     which is dead by then (the next DMA into it is issued two barriers later).
#include "rn_conv_big_epi.h"

// HALO_ABLATE (probe builds only, tools/probes/build_halo_ablate.sh): 1 = no counted vmcnt wait, 2 = no DMA issue,
// 4 = no fragment reads, 8 = no MFMAs — wrong results, timing only
#ifndef HALO_ABLATE
#define HALO_ABLATE 0
#endif

#ifdef HALO_PROF   // probe builds: core clock (clock64) and 100 MHz wall clock at the start / end of workgroup 0
extern "C" __attribute__((visibility("default"))) int rn_debug_halo_clocks(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_halo_clk), sizeof(g_halo_clk)) == hipSuccess ? 0 : -3;
}
#endif

namespace {

constexpr int BK = 32, NW = 8;
constexpr int W_STAGES = 3;                              // 9 taps = 3 x 3: the ring stage of a tap is tap % 3
// Tile geometry by wave grid.  The eight waves always own 128 pixels x 64 channels each (4 x 2 MFMA tiles); WM of them
// stack along the pixels, 8 / WM along the channels:
//   WM = 2: 256 pixels x 256 channels — every layer with Cout >= 256 (the round-1..3 kernel);
//   WM = 4: 512 pixels x 128 channels — 3x3 layers with 64 < Cout <= 128 (ResNet stage 2, 128 -> 128 at 80 x 80): in the
//           256-wide tile half of the waves multiplied zero weight rows, and on the 128-row kernel the layer re-staged
//           its pixels once per tap (round 4).
template <int WM>
struct HaloGeo {
  static constexpr int WN = 8 / WM;
  static constexpr int BM = 128 * WM, BN = 64 * WN;
  static constexpr int PIX_PX = WM == 2 ? 640 : 896;       // patch capacity in pixels (a multiple of 64): a 512-pixel tile of an
                                                           // 80 x 80 level that straddles two images needs 11 rows x 81 + 1 = 892
  static constexpr int PIX_BYTES = PIX_PX * 64;            // one halo patch (40 / 56 KB)
  static constexpr int BLOCKS = PIX_PX / 64;               // 64-pixel blocks per plane
  static constexpr int PIECES = (BLOCKS * 4 + NW - 1) / NW;   // patch DMA pieces per wave per chunk: 5 / 7
  static constexpr int PLANE = PIX_PX * 16;                // the patch is 4 planes [16-byte channel slot][pixel]
  static constexpr int W_STAGE = BN * BK * 2;              // 16 / 8 KB of weights per K step
  static constexpr int W_PIECES = W_STAGE / 1024 / NW;     // weight DMA pieces per wave per K step: 2 / 1
  static constexpr int W_RING = 2 * PIX_BYTES;
  static constexpr int PA_TABLE = W_RING + W_STAGES * W_STAGE;   // source offset of every patch pixel (plane 0), one dword each
  static constexpr int BASE_TABLE = PA_TABLE + PIX_PX * 4;       // 2 x BM dwords: patch offset of every tile pixel, a ring of two tiles
  static constexpr int SEG_TABLE = BASE_TABLE + 2 * BM * 4;      // segment descriptors the tile set-ups need (see HaloSeg)
  static constexpr int LDS_BYTES = SEG_TABLE + 16 * 4 + RN_CONV_MAX_SEGMENTS * 64;   // 133 / 148 KB
  static_assert(PIECES <= 8, "one patch piece per load segment of taps 1..8");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// What the three tile set-ups read of a segment, copied into LDS once per kernel: reading ConvArgs through the
// scalar cache cost each set-up a chain of dependent s_loads (the segment search alone one per segment) — measured
// ~5 000 cycles per set-up, three set-ups per tile.  From LDS it is one round of broadcast reads.
struct HaloSeg {
  const uint16_t* x;
  const uint16_t* w;
  int N, H, W, pix_stride, Cout, M, tile_begin, n_tiles, CinP, cwrap, pitch;
};
// layout: 16 dwords of tile_begin (INT_MAX past the last segment), then 16 dwords per segment
template <int SEG_TABLE>
__device__ __forceinline__ void halo_seg_table_fill(char* smem, const ConvArgs& args, int tid) {
  int* tb = (int*)(smem + SEG_TABLE);
  if (tid < 16) tb[tid] = tid < args.nseg ? args.seg[tid < RN_CONV_MAX_SEGMENTS ? tid : 0].tile_begin : 0x7fffffff;
  if (tid < args.nseg) {
    const ConvSegDev& g = args.seg[tid];
    unsigned* d = (unsigned*)(smem + SEG_TABLE + 64 + tid * 64);
    const unsigned long long px = (unsigned long long)g.x, pw = (unsigned long long)g.w;
    d[0] = (unsigned)px; d[1] = (unsigned)(px >> 32); d[2] = (unsigned)pw; d[3] = (unsigned)(pw >> 32);
    d[4] = g.N; d[5] = g.H; d[6] = g.W; d[7] = g.pix_stride;
    d[8] = g.Cout; d[9] = g.M; d[10] = g.tile_begin; d[11] = g.n_tiles;
    d[12] = g.CinP; d[13] = g.cwrap; d[14] = g.halo_pitch; d[15] = 0;
  }
}
template <int SEG_TABLE>
__device__ __forceinline__ int halo_seg_of_tile(const char* smem, int tile) {
  const int4* tb = (const int4*)(smem + SEG_TABLE);
  const int4 a = tb[0], b = tb[1], c = tb[2];
  int si = (tile >= a.y) + (tile >= a.z) + (tile >= a.w) + (tile >= b.x) + (tile >= b.y) + (tile >= b.z) +
           (tile >= b.w) + (tile >= c.x) + (tile >= c.y);
  return __builtin_amdgcn_readfirstlane(si);
}
template <int SEG_TABLE>
__device__ __forceinline__ HaloSeg halo_seg(const char* smem, int si) {
  const uint4* d = (const uint4*)(smem + SEG_TABLE + 64 + si * 64);
  const uint4 q0 = d[0], q1 = d[1], q2 = d[2], q3 = d[3];
  HaloSeg g;
#define HALO_U(v_) ((unsigned)__builtin_amdgcn_readfirstlane((int)(v_)))
  g.x = (const uint16_t*)(((unsigned long long)HALO_U(q0.y) << 32) | HALO_U(q0.x));
  g.w = (const uint16_t*)(((unsigned long long)HALO_U(q0.w) << 32) | HALO_U(q0.z));
  g.N = (int)HALO_U(q1.x); g.H = (int)HALO_U(q1.y); g.W = (int)HALO_U(q1.z); g.pix_stride = (int)HALO_U(q1.w);
  g.Cout = (int)HALO_U(q2.x); g.M = (int)HALO_U(q2.y); g.tile_begin = (int)HALO_U(q2.z); g.n_tiles = (int)HALO_U(q2.w);
  g.CinP = (int)HALO_U(q3.x); g.cwrap = (int)HALO_U(q3.y); g.pitch = (int)HALO_U(q3.z);
#undef HALO_U
  return g;
}

// Work unit of a virtual id (rn_conv_dev.h, ConvArgs::split_f): a whole tile, or part `part` of `nparts` of a tile of the
// launch's last round — the channel chunks [part * nch / nparts, (part + 1) * nch / nparts) of it.
// SPLIT launches (the split last round, rn_conv_dev.h): work unit v = l * split_s + part is part `part` of tile
// split_f + l — the channel chunks [part * nch / split_s, (part + 1) * nch / split_s) of it.  The whole-tile launches
// (SPLIT = false) compile to the round-3 code: unit = tile tile_of(v, total_tiles), every chunk.
struct HaloUnit { int tile, begin, end; };
template <bool SPLIT>
__device__ __forceinline__ HaloUnit halo_unit(int v, const ConvArgs& a, int total, int nch) {
  HaloUnit u;
  if (!SPLIT) {
    u.tile = tile_of(v, total); u.begin = 0; u.end = nch;
  } else {
    const int S = a.split_s;
    const float rS = rn_rcp((float)S);
    const int l = rn_fdiv(v, S, rS), part = v - l * S;
    u.tile = a.split_f + l;
    u.begin = rn_fdiv(part * nch, S, rS);
    u.end = rn_fdiv((part + 1) * nch, S, rS);
  }
  return u;
}
template <bool SPLIT>
__device__ __forceinline__ int halo_tile_of(int v, const ConvArgs& a, int total) {
  if (!SPLIT) return tile_of(v, total);
  return a.split_f + rn_fdiv(v, a.split_s, rn_rcp((float)a.split_s));
}

// ---- a part of a split tile (rnet_hip.h: rn_conv_problem.splitk_ws) ------------------------------------------------
// EVERY part stores its raw fp32 accumulators (lane-linear, 16 B per lane and store) into its slot and counts itself in;
// whoever arrives LAST runs the tile's epilogue, which REBUILDS the accumulators from the slots, one 32-pixel block at a
// time, parts added in PART order (deterministic whoever the last arriver is) — the own slot included, so that the
// registers the K loop accumulated in are dead after the stores (big_epilogue<.., FROM_WS>): `acc += slot` on the live
// registers, and all eight blocks rebuilt up front, both made the allocator spill 130 - 350 bytes per lane, and a kernel
// with scratch pays ~40 us per LAUNCH on this stack (the runtime attaches scratch memory per dispatch: measured,
// gpurun_out/r04c).  Nobody waits for anybody (round 4: part 0 polled the counter, bounded by a 2 s timeout whose exit
// left the counter in a state that poisoned later launches — ADVICE r4), so there is no timeout and no status word.
// The hand-off is PER WAVE: a slot is [wave][32 x (64 lanes x 16 B)] and the epilogue of wave w reads only the w-th
// eighth of every part's slot, so wave w of each part counts itself in on counter (tile, w) and the last wave w to arrive
// — of whichever part — finishes that eighth of the tile; no workgroup barrier.  It uses no cache-wide operation: the
// slots are written with WRITE-THROUGH stores (sc1) and read back with sc1 loads (cdna_hip_programming.md section 6,
// guideline 16, R1): the storing wave drains its stores (vmcnt 0), ONE lane bumps the counter with a returning agent-scope
// atomic, and the value it gets back (S - 1: every other part's wave w has drained and counted) gates the sc1 loads.
// (The first version released / acquired at agent scope: buffer_wbl2 writes back the whole L2 of the XCD — the previous
// launch's output tiles — and buffer_inv drops it, weights included, for every workgroup on that XCD: +30 us per launch.)
// The last arriver zeroes the counter: the header is all zero again when the launch completes.  Returns the tile's first
// slot (l * S) to the wave that runs the epilogue, -1 to the others.
// One slot = 256 KB; buffer addressing: ONE lane offset register, the piece is a scalar offset.
typedef unsigned halo_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void halo_split_store(const f32x16_t (&acc)[4][2], const ConvArgs& args, int slot, int voff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (char*)args.ws + RN_SPLITK_HEADER_BYTES + (size_t)slot * RN_SPLITK_SLOT_BYTES, 0, RN_SPLITK_SLOT_BYTES, 0x00020000);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        halo_u32x4_t v = {__float_as_uint(acc[i][j][q * 4]), __float_as_uint(acc[i][j][q * 4 + 1]),
                          __float_as_uint(acc[i][j][q * 4 + 2]), __float_as_uint(acc[i][j][q * 4 + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, ((i * 2 + j) * 4 + q) * 1024, 16);   // aux 16 = sc1: write-through
      }
}
__device__ __forceinline__ int halo_split_exchange(f32x16_t (&acc)[4][2], const ConvArgs& args, int c_v, int wave) {
  const int S = args.split_s;
  const int l = c_v / S;     // leftover tile index (scalar division, once)
  const int part = c_v - l * S;
  unsigned* const cnt = (unsigned*)args.ws + l * 8 + wave;   // header: one counter per (tile, wave)
  const unsigned lane_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int voff = (int)(wave * 32768 + lane_ * 16);
  halo_split_store(acc, args, l * S + part, voff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned old = 0;
  if (lane_ == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if (old != (unsigned)(S - 1)) return -1;
  if (lane_ == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  asm volatile("" ::: "memory");
  return l * S;
}

template <bool OUT_F32, bool HAS_RES, bool BN_BWD = false, bool SPLIT = false, int WM = 2>   // BN_BWD: rn_conv_big_epi.h
__global__ void __launch_bounds__(512) conv_halo_kernel(const ConvArgs args) {
  using GE = HaloGeo<WM>;
  constexpr int WN = GE::WN, BM = GE::BM, BN = GE::BN, PIX_PX = GE::PIX_PX, PIX_BYTES = GE::PIX_BYTES, BLOCKS = GE::BLOCKS;
  constexpr int PIECES = GE::PIECES, PLANE = GE::PLANE, W_STAGE = GE::W_STAGE, W_PIECES = GE::W_PIECES, W_RING = GE::W_RING;
  constexpr int PA_TABLE = GE::PA_TABLE, BASE_TABLE = GE::BASE_TABLE, SEG_TABLE = GE::SEG_TABLE;
  static_assert(!SPLIT || WM == 2, "the split last round exists for the 256 x 256 tiles only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total = SPLIT ? args.vtotal : args.total_tiles;   // work units: tiles, or the parts of the split last round
  const int G = gridDim.x;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int grp = wave >> 2;   // ping-pong group: waves w and w + 4 share a SIMD (WM = 2: the group is wave_m)

  // ---- issue side, pixels: the chunk whose patch is being DMA'd (one chunk ahead of the weights' chunk) ----
#ifdef HALO_PROF
  int stamp_n_[4] = {0, 0, 0, 0};
#endif
  int p_v = blockIdx.x;        // virtual tile id; exhausted when >= total
  int p_chunk = 0, p_nch = 0;  // chunk of that unit / the chunk its range ends at
  int p_wrap = 0;              // input channel chunks before they repeat (split-bf16 weight planes)
  int p_par = 0;               // patch buffer it goes to; bit 1: parity of the tile's ordinal (ring slot of its base table)
  __amdgpu_buffer_rsrc_t rs_x;
  // PA table (LDS, shared by the workgroup): byte offset of patch pixel p's first 16 bytes at chunk 0, or RN_OOB — one
  // dword per patch pixel.  The four waves that stage the four 16-byte planes of a pixel used to compute the same
  // (image, row, column) each: 5 entries per thread and tile, ~2 500 cycles of VALU in one load segment — and a long load
  // segment stalls the other group at the barrier, so the two groups' set-ups cost a tile ~9 % of its time (DESIGN.md
  // section 4).  Now every thread computes ONE patch pixel (the first two waves a second one: 640 pixels) and, waves
  // 4..7, one entry of the base table the compute side reads at the tile's start (the patch offset of a tile pixel: each
  // was computed by 8 lanes).  Written in the load segment of tap 5, first read two barriers later at the earliest.
  // The table address is rebuilt from a fresh lane id at every use (volatile: not hoisted), so that no
  // per-thread address stays live across the epilogue — the allocator would spill it and reload it, with an
  // `s_waitcnt vmcnt(0)`, in every load segment.
#ifdef HALO_PROF   // clock stamps around the first in-loop pixel / weight set-up of workgroup 0, wave 0
#define HALO_SETUP_STAMP(k_) if (blockIdx.x == 0 && tid == 0 && stamp_n_[(k_) - 32]++ == 0) g_halo_clk[k_] = clock64();
#else
#define HALO_SETUP_STAMP(k_)
#endif
#define HALO_LANE(dst_) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(dst_))
// entry of the pixel this lane stages in piece j: pixel block 2j + wave/4, one pixel per lane
#define HALO_PA_AT(j_, ln_) (*(unsigned*)(smem + PA_TABLE + ((2 * (j_) + (wave >> 2)) * 64 + (int)(ln_)) * 4))
#define HALO_PA_PX(p_) (*(unsigned*)(smem + PA_TABLE + (p_) * 4))
#define HALO_BASE_AT(slot_, m_) (*(int*)(smem + BASE_TABLE + (slot_) * (BM * 4) + (m_) * 4))

#define HALO_SETUP_PIX()                                                                              \
  do {                                                                                                \
    const int tile__ = halo_tile_of<SPLIT>(p_v, args, total);                                         \
    const int si__ = halo_seg_of_tile<SEG_TABLE>(smem, tile__);                                                  \
    const HaloSeg sg__ = halo_seg<SEG_TABLE>(smem, si__);                                                        \
    const int lt__ = tile__ - sg__.tile_begin;                                                        \
    const int m0__ = rn_fdiv(lt__, sg__.n_tiles, rn_rcp((float)sg__.n_tiles)) * BM;                \
    const int H__ = sg__.H, W__ = sg__.W, PS__ = sg__.pix_stride, W1__ = sg__.pitch, H1__ = H__ + 1;  \
    const int HW__ = H__ * W__;                                                                       \
    {                                                                                                 \
      const HaloUnit un__ = halo_unit<SPLIT>(p_v, args, total, sg__.CinP / BK);                       \
      p_chunk = un__.begin;                                                                           \
      p_nch = un__.end;                                                                               \
    }                                                                                                 \
    p_wrap = sg__.cwrap / BK;                                                                         \
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)sg__.x, 0,                                        \
                                             (int)((long long)sg__.N * HW__ * PS__ * 2), 0x00020000); \
    const int ml__ = (m0__ + BM - 1 < sg__.M ? m0__ + BM - 1 : sg__.M - 1);                           \
    const float rHW__ = rn_rcp((float)HW__), rW__ = rn_rcp((float)W__);                         \
    const float rW1__ = rn_rcp((float)W1__), rH1__ = rn_rcp((float)H1__);                       \
    const int nf__ = rn_fdiv(m0__, HW__, rHW__), nl__ = rn_fdiv(ml__, HW__, rHW__);                   \
    const int Gf__ = nf__ * H1__ + rn_fdiv(m0__ - nf__ * HW__, W__, rW__) + 1;   /* padded row of the first pixel */ \
    const int Gl__ = nl__ * H1__ + rn_fdiv(ml__ - nl__ * HW__, W__, rW__) + 1;                        \
    const int plast__ = (Gl__ - Gf__ + 3) * W1__;   /* last patch pixel: the zero right of the last row */ \
    unsigned ln__;                                                                                    \
    HALO_LANE(ln__);                                                                                  \
    const int t__ = wave * 64 + (int)ln__;                                                            \
    /* patch pixel t (and 512 + t for the first two waves): plane 0's byte offset; piece j of wave w stages plane */ \
    /* (w & 3) of pixel block 2j + (w >> 2) and adds its plane's 16 bytes at issue */                  \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                   \
      if (k == 1 && wave >= (PIX_PX - 512) / 64) break;                                               \
      const int p = t__ + k * 512;                                                                    \
      const int prow = rn_fdiv(p, W1__, rW1__), pcol = p - prow * W1__;                               \
      const int Gp = Gf__ - 1 + prow;                                                                 \
      const int n = rn_fdiv(Gp, H1__, rH1__);                                                         \
      const int iy = Gp - n * H1__ - 1, ix = pcol - 1;                                                \
      const bool ok = (unsigned)iy < (unsigned)H__ && (unsigned)ix < (unsigned)W__ && p < plast__ &&  \
                      n < sg__.N;                                                                     \
      HALO_PA_PX(p) = ok ? (unsigned)((((n * H__ + iy) * W__ + ix) * PS__) * 2) : RN_OOB;   /* < 2 GiB: checked by the host */ \
    }                                                                                                 \
    if (WM == 4 || wave >= 4) {   /* base table of this tile (ring slot = ordinal parity): one tile pixel per thread */ \
      const int ml0__ = WM == 4 ? t__ : t__ - 256;   /* 256-pixel tiles: the second half of the threads */  \
      int m = m0__ + ml0__;                                                                           \
      m = m < sg__.M ? m : sg__.M - 1;                                                                \
      const int n = rn_fdiv(m, HW__, rHW__);                                                          \
      const int rem = m - n * HW__;                                                                   \
      const int oy = rn_fdiv(rem, W__, rW__), ox = rem - oy * W__;                                    \
      HALO_BASE_AT((p_par >> 1) & 1, ml0__) = ((n * H1__ + oy + 1 - Gf__) * W1__ + ox) * 16;          \
    }                                                                                                 \
    p_par ^= 2;                                                                                       \
  } while (0)

// piece j of this wave = pixel block 2j + wave/4: past the patch for the last piece of waves 4..7 when the block count is odd
#define HALO_PIECE_LIVE(j_) (2 * (j_) + 1 < BLOCKS || 2 * (j_) + (wave >> 2) < BLOCKS)
// one piece of the next chunk's patch (piece index compile time); after the last piece the stream advances
#define HALO_ISSUE_PIX(j_, pa_)                                                                       \
  do {                                                                                                \
    const int pc__ = p_chunk < p_wrap ? p_chunk : (p_chunk < 2 * p_wrap ? p_chunk - p_wrap : p_chunk - 2 * p_wrap); \
    const unsigned v__ = (pa_) == RN_OOB ? RN_OOB : (pa_) + (unsigned)(pc__ * (BK * 2) + (wave & 3) * 16); \
    if (!(HALO_ABLATE & 2) && HALO_PIECE_LIVE(j_))                                                    \
      dma16(rs_x, smem + (p_par & 1) * PIX_BYTES + (wave & 3) * PLANE + (2 * (j_) + (wave >> 2)) * 1024, v__); \
    if ((j_) == PIECES - 1 && p_v < total) {                                                          \
      p_par ^= 1;                                                                                     \
      if (++p_chunk == p_nch) {                                                                       \
        p_v = SPLIT ? total : p_v + G;   /* SPLIT: one unit per workgroup, no next-tile code */       \
        if (!SPLIT && p_v < total) {                                                                  \
          HALO_SETUP_STAMP(32);                                                                       \
          HALO_SETUP_PIX();                                                                           \
          HALO_SETUP_STAMP(33);                                                                       \
        } else {  /* end of the stream: the remaining pieces are zero fills into the dead buffer */   \
          unsigned ln2__;                                                                             \
          HALO_LANE(ln2__);                                                                           \
          HALO_PA_PX(wave * 64 + (int)ln2__) = RN_OOB;                                                \
          if (wave < (PIX_PX - 512) / 64) HALO_PA_PX(512 + wave * 64 + (int)ln2__) = RN_OOB;          \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
  } while (0)

  // ---- issue side, weights: the tile / chunk / tap whose 16 KB are being DMA'd (3 steps ahead) -------------
  int w_v = blockIdx.x;
  int w_chunk = 0, w_nch = 0, w_end = 0;   // chunk, chunks of the whole tile (the K stride of a tap), end of the unit's range
  __amdgpu_buffer_rsrc_t rs_w;
  unsigned b_off;              // piece 0 (rows wave*16 + lane/4 of the tile); piece 1 is 128 rows further
  unsigned w_step1 = 0;        // byte distance of piece 1, or RN_OOB when those rows are past the packed weights
  // (weight piece j fills rows (j*8 + wave)*16 + lane/4 of the stage, 16-byte slot lane%4)

#define HALO_SETUP_W()                                                                                \
  do {                                                                                                \
    const int tile__ = halo_tile_of<SPLIT>(w_v, args, total);                                         \
    const int si__ = halo_seg_of_tile<SEG_TABLE>(smem, tile__);                                                  \
    const HaloSeg sg__ = halo_seg<SEG_TABLE>(smem, si__);                                                        \
    const int lt__ = tile__ - sg__.tile_begin;                                                        \
    const int n0__ = (lt__ - rn_fdiv(lt__, sg__.n_tiles, rn_rcp((float)sg__.n_tiles)) * sg__.n_tiles) * BN; \
    w_nch = sg__.CinP / BK;                                                                           \
    {                                                                                                 \
      const HaloUnit un__ = halo_unit<SPLIT>(w_v, args, total, w_nch);                                \
      w_chunk = un__.begin;                                                                           \
      if (SPLIT) w_end = un__.end;                                                                    \
    }                                                                                                 \
    const int Ktot__ = 9 * sg__.CinP;                                                                 \
    const int rows__ = sg__.Cout <= 64 ? 64 : ((sg__.Cout + 127) / 128) * 128; /* packed weight rows (rn_conv_cout_pad) */ \
    rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)sg__.w, 0, (int)((long long)rows__ * Ktot__ * 2), \
                                             0x00020000);                                             \
    {                                                                                                 \
      unsigned lw__;                                                                                  \
      HALO_LANE(lw__);                                                                                \
      const int row = wave * 16 + (int)(lw__ >> 2);   /* piece 1: row + 128, same swizzle */           \
      const int chunk = (int)(lw__ & 3) ^ lds_swz<BK>(row);                                           \
      b_off = (unsigned)(((long long)(n0__ + row) * Ktot__ + chunk * 8) * 2);                         \
      /* rows__ is a multiple of 128 and n0 < rows__: piece 0 is always inside, piece 1 as a whole or not */ \
      w_step1 = n0__ + 128 < rows__ ? (unsigned)(128 * Ktot__ * 2) : RN_OOB;                          \
    }                                                                                                 \
  } while (0)

// the two weight pieces of the next stream step, tap compile time
#define HALO_ISSUE_W(tap_)                                                                            \
  do {                                                                                                \
    const unsigned koff__ = (unsigned)(((tap_) * w_nch + w_chunk) * (BK * 2));                        \
    char* st__ = smem + W_RING + ((tap_) % 3) * W_STAGE;                                              \
    if (!(HALO_ABLATE & 2)) {                                                                         \
      dma16(rs_w, st__ + wave * 1024, b_off + koff__);                                                \
      if (W_PIECES == 2)   /* a 256-row stage; the 128-row stage of the 512 x 128 tiles is one piece per wave */ \
        dma16(rs_w, st__ + (NW + wave) * 1024, w_step1 == RN_OOB ? RN_OOB : b_off + koff__ + w_step1); \
    }                                                                                                 \
    if ((tap_) == 8) {                                                                                \
      if (++w_chunk == (SPLIT ? w_end : w_nch)) {                                                     \
        w_v = SPLIT ? total : w_v + G;                                                                \
        if (!SPLIT && w_v < total) { HALO_SETUP_STAMP(34); HALO_SETUP_W(); HALO_SETUP_STAMP(35); }    \
      }                                                                                               \
    }                                                                                                 \
  } while (0)

  // ---- compute side: the tile being accumulated ---------------------------------------------------------
  int c_v = blockIdx.x;
  int c_chunk = 0, c_nch = 0, c_par = 0;   // c_nch: the chunk the unit's range ends at
  int c_m0 = 0, c_n0 = 0, c_si = 0, c_W1 = 0;
  int base[4];   // LDS byte offset (inside a patch) of this lane's 16 bytes of its 4 fragment rows at tap (0,0)
  // weights: row = lane&31 of a 32-row tile, 16-byte slot = 2*kk + (lane>>5), XOR-swizzled by (row/4)&3.
  // Rebuilt from a fresh lane id after every epilogue so that it is not live (and spilled) across it.
  int off_w0;
#define HALO_LANE_CONSTS()                                                                            \
  do {                                                                                                \
    unsigned lc__;                                                                                    \
    HALO_LANE(lc__);                                                                                  \
    const int fr__ = (int)(lc__ & 31), fh__ = (int)(lc__ >> 5);                                       \
    off_w0 = (wave_n * 64 + fr__) * 64 + ((fh__ ^ ((fr__ >> 2) & 3)) << 4);                           \
  } while (0)
  HALO_LANE_CONSTS();

#define HALO_SETUP_COMPUTE()                                                                          \
  do {                                                                                                \
    const int tile__ = halo_tile_of<SPLIT>(c_v, args, total);                                         \
    c_si = halo_seg_of_tile<SEG_TABLE>(smem, tile__);                                                            \
    const HaloSeg sg__ = halo_seg<SEG_TABLE>(smem, c_si);                                                        \
    const int lt__ = tile__ - sg__.tile_begin;                                                        \
    const int mt__ = rn_fdiv(lt__, sg__.n_tiles, rn_rcp((float)sg__.n_tiles));                     \
    c_m0 = mt__ * BM;                                                                                 \
    c_n0 = (lt__ - mt__ * sg__.n_tiles) * BN;                                                         \
    {                                                                                                 \
      const HaloUnit un__ = halo_unit<SPLIT>(c_v, args, total, sg__.CinP / BK);                       \
      c_chunk = un__.begin;                                                                           \
      c_nch = un__.end;                                                                               \
    }                                                                                                 \
    c_W1 = sg__.pitch;                                                                                \
    unsigned lq__;                                                                                    \
    HALO_LANE(lq__);                                                                                  \
    const int fr = (int)(lq__ & 31), fh = (int)(lq__ >> 5);                                           \
    /* the patch offsets of this lane's four fragment pixels: written by the pixel stream's set-up of this tile (a */ \
    /* pass or more ago, barriers in between), ring slot = parity of the tile's ordinal */             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
      base[i] = HALO_BASE_AT((c_par >> 1) & 1, wave_m * 128 + i * 32 + fr) + fh * PLANE;              \
    c_par ^= 2;                                                                                       \
  } while (0)

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  bf16x8_t px0[4], wt0[2], px1[4], wt1[2];   // fragments of one stream step: K slices 0..15 / 16..31
  unsigned pa_next = RN_OOB;                  // source offset of the patch piece the next load segment issues
// fragments of the step with tap `tap_` (compile time): weights from ring stage tap % 3, pixels from the current
// patch: plane (2*kk + lane/32), pixel base + r*(W+1) + s — one VALU add per fragment, kk = 1 is an immediate
#define HALO_READ(tap_)                                                                               \
  do {                                                                                                \
    const char* wb__ = smem + W_RING + ((tap_) % 3) * W_STAGE;                                        \
    const char* pb__ = smem + (c_par & 1) * PIX_BYTES + (((tap_) / 3) * c_W1 + ((tap_) % 3)) * 16;    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) wt0[j] = *(const bf16x8_t*)(wb__ + off_w0 + j * 2048); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) px0[i] = *(const bf16x8_t*)(pb__ + base[i]);        \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) wt1[j] = *(const bf16x8_t*)(wb__ + (off_w0 ^ 32) + j * 2048); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) px1[i] = *(const bf16x8_t*)(pb__ + base[i] + 2 * PLANE); \
  } while (0)
// load segment for the step with tap `tap_`: fragment reads first (their LDS latency runs under the DMA issue),
// then this wave's pieces — weights of the step 2 ahead and, in taps 1..5, one piece of the next chunk's patch
// (its source offset was fetched from the LDS table one segment earlier) — then the counted wait: everything
// but this segment's own pieces
#define HALO_LOADSEG(tap_)                                                                            \
  do {                                                                                                \
    if (!(HALO_ABLATE & 4)) HALO_READ(tap_);                                                          \
    constexpr bool pix__ = (tap_) >= 1 && (tap_) <= PIECES;                                           \
    if (w_v < total) {                                                                                \
      HALO_ISSUE_W(((tap_) + 2) % 9);                                                                 \
      if (pix__) HALO_ISSUE_PIX((tap_) - 1, pa_next);                                                 \
      /* the counted wait: everything but this segment's own pieces (W_PIECES of weights + a live patch piece) */ \
      if (HALO_ABLATE & 3) {                                                                          \
      } else if (pix__ && HALO_PIECE_LIVE((tap_) - 1)) {                                              \
        if (W_PIECES == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                           \
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                         \
      } else {                                                                                        \
        if (W_PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                           \
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                         \
      }                                                                                               \
    } else {                                                                                          \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
    }                                                                                                 \
    if ((tap_) < PIECES) {   /* the offset the next segment's patch piece needs */                    \
      unsigned ln__;                                                                                  \
      HALO_LANE(ln__);                                                                                \
      pa_next = HALO_PA_AT((tap_), ln__);                                                             \
    }                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
  } while (0)
#define HALO_COMPUTESEG()                                                                             \
  do {                                                                                                \
    if (HALO_ABLATE & 8) break;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        acc[i][j] = RN_MFMA_32x32x16(wt0[j], px0[i], acc[i][j], 0, 0, 0);      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        acc[i][j] = RN_MFMA_32x32x16(wt1[j], px1[i], acc[i][j], 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#define HALO_BARRIER()                      \
  do {                                      \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

#ifdef HALO_PROF
  if (blockIdx.x == 0 && tid == 0) { g_halo_clk[0] = clock64(); g_halo_clk[1] = wall_clock64(); for (int q = 8; q < 48; ++q) g_halo_clk[q] = 0; }
#endif
  // ---- prologue -------------------------------------------------------------------------------------------
  halo_seg_table_fill<SEG_TABLE>(smem, args, tid);
  __syncthreads();
  HALO_SETUP_PIX();
  HALO_SETUP_W();
  __syncthreads();   // the PA / base tables are written by all waves
  HALO_SETUP_COMPUTE();
  big_acc_init<OUT_F32, HAS_RES, WM>(acc, args, c_si, c_n0, wave, !SPLIT || c_chunk == 0);   // the bias belongs to part 0 of a split tile
  // patch of the unit's first chunk (5 pieces), then the weights of stream steps 0 and 1
  {
    unsigned pa0[PIECES];
    unsigned ln0;
    HALO_LANE(ln0);
#pragma unroll
    for (int j = 0; j < PIECES; ++j) pa0[j] = HALO_PA_AT(j, ln0);
    __syncthreads();   // a one-chunk tile: the last piece below already sets up the NEXT tile's tables
    HALO_ISSUE_PIX(0, pa0[0]); HALO_ISSUE_PIX(1, pa0[1]); HALO_ISSUE_PIX(2, pa0[2]); HALO_ISSUE_PIX(3, pa0[3]);
    HALO_ISSUE_PIX(4, pa0[4]);
    if (PIECES == 7) { HALO_ISSUE_PIX(5, pa0[PIECES == 7 ? 5 : 0]); HALO_ISSUE_PIX(6, pa0[PIECES == 7 ? 6 : 0]); }
    static_assert(PIECES == 5 || PIECES == 7, "prologue issues the first chunk's pieces by hand");
  }
  HALO_ISSUE_W(0); HALO_ISSUE_W(1);
  // the patch and step 0 have landed (step 1's weight pieces may still fly)
  if (W_PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  HALO_BARRIER();
  // pre-roll: both groups read step 0 and issue step 2; group 1 does so as its slot-0 load segment
  HALO_LOADSEG(0);
  if (grp == 1) HALO_BARRIER();

  // ---- main loop: nine taps per pass, one loop body for both groups --------------------------------------------
  // slot 2g: group 0 compute(g) | group 1 load(read g, issue g+3); slot 2g+1: group 0 load(read g+1, issue g+4)
  // | group 1 compute(g).  A finished tile is written out in the odd slot by both groups (group 1 right after
  // its MFMAs, group 0 before its load segment).
#define HALO_CHUNK_END()                                                                  \
  c_par ^= 1;                                                                             \
  if (__builtin_expect(++c_chunk == c_nch, 0)) {                                          \
    if (SPLIT) break;   /* a part of a split tile, the workgroup's only unit: handled behind the loop */ \
    HALO_EPI_PROBE(8);                                                                    \
    big_epilogue<OUT_F32, HAS_RES, BN_BWD, false, WM>(acc, args, c_si, c_m0, c_n0, wave,                   \
                          smem + ((c_par ^ 1) & 1) * PIX_BYTES + wave * 4096 /* 32 KB of the dead 40 KB patch */); \
    HALO_EPI_PROBE(9);                                                                    \
    HALO_EPI_COUNT();                                                                     \
    if (c_v + G >= total) break;                                                          \
    c_v += G;                                                                             \
    HALO_LANE_CONSTS();                                                                   \
    HALO_SETUP_COMPUTE();                                                                 \
    big_acc_init<OUT_F32, HAS_RES, WM>(acc, args, c_si, c_n0, wave, !SPLIT || c_chunk == 0);  \
    HALO_EPI_PROBE_AT(11, 2);                                                             \
  }
// One loop body for both groups (the barrier that follows the compute segment sits before the tile-end work for
// group 0 and after it for group 1): two uniform branches per K step, but half the code — the kernel is larger
// than the instruction cache, and every tile's epilogue / set-up code is fetched again.
#define HALO_STEP(next_tap_)                     \
  HALO_COMPUTESEG();                             \
  if (grp == 0) HALO_BARRIER();               \
  if ((next_tap_) == 0) { HALO_CHUNK_END() }     \
  if (grp == 1) HALO_BARRIER();               \
  HALO_LOADSEG(next_tap_);                       \
  if ((next_tap_) == 0) { HALO_EPI_PROBE2() }    \
  HALO_BARRIER();
#ifdef HALO_PROF
#define HALO_EPI_COUNT() ++epi_
#define HALO_EPI_PROBE_AT(k_, e_) if (blockIdx.x == 0 && tid == 0 && epi_ == (e_)) g_halo_clk[k_] = clock64();
#define HALO_EPI_PROBE(k_) if (blockIdx.x == 0 && tid == 0 && epi_ == 1) g_halo_clk[k_] = clock64();
#else
#define HALO_EPI_COUNT()
#define HALO_EPI_PROBE_AT(k_, e_)
#define HALO_EPI_PROBE(k_)
#endif
#ifdef HALO_PROF
#define HALO_EPI_PROBE2() if (blockIdx.x == 0 && tid == 0 && epi_ == 2) g_halo_clk[10] = clock64();
#else
#define HALO_EPI_PROBE2()
#endif
#ifdef HALO_PROF
  int g_pass_after_ = 0;
  int pass_ = 0, epi_ = 0;   // stamps of the SECOND tile of workgroup 0
#define HALO_PASS_PROBE() \
  if (blockIdx.x == 0 && tid == 0 && pass_ >= 2 && pass_ < 6) g_halo_clk[2 + pass_] = clock64(); \
  if (blockIdx.x == 0 && tid == 0 && pass_ < 10) g_halo_clk[22 + pass_] = clock64(); \
  ++pass_;
#else
#define HALO_PASS_PROBE()
#endif
#pragma unroll 1
  while (true) {
    HALO_PASS_PROBE()
    HALO_STEP(1) HALO_STEP(2) HALO_STEP(3) HALO_STEP(4) HALO_STEP(5) HALO_STEP(6) HALO_STEP(7)
    HALO_STEP(8) HALO_STEP(0)
  }
  if (SPLIT) {   // partial accumulators through the workspace; per wave, the last part to arrive runs the epilogue
    if (grp == 1) HALO_BARRIER();   // group 0 left the loop one barrier ahead of group 1: every wave's MFMAs are done
    const int slot0 = halo_split_exchange(acc, args, c_v, wave);
    if (slot0 >= 0) {
      BigEpiSrc src;
      unsigned lane2_;
      HALO_LANE(lane2_);
      src.slots = (const char*)args.ws + RN_SPLITK_HEADER_BYTES + (size_t)slot0 * RN_SPLITK_SLOT_BYTES;
      src.nparts = args.split_s;
      src.voff = (int)(wave * 32768 + lane2_ * 16);
      big_epilogue<OUT_F32, HAS_RES, BN_BWD, true, WM>(acc, args, c_si, c_m0, c_n0, wave,
                                                   smem + ((c_par ^ 1) & 1) * PIX_BYTES + wave * 4096, src);
    }
  }
#ifdef HALO_PROF
  if (blockIdx.x == 0 && tid == 0) { g_halo_clk[2] = clock64(); g_halo_clk[3] = wall_clock64(); }
#endif
}

}  // namespace

template <bool SPLIT, int WM>
static int halo_launch(const ConvArgs& a, bool out_f32, int grid, hipStream_t st) {
  constexpr int LDS = HaloGeo<WM>::LDS_BYTES;
  static unsigned long long attr_set = 0;   // per template instantiation, one bit per device
  if (RN_ATTRS_NEEDED(attr_set)) {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<false, false, false, SPLIT, WM>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<false, true, false, SPLIT, WM>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<true, false, false, SPLIT, WM>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<true, true, false, SPLIT, WM>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo_kernel<false, false, true, SPLIT, WM>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    RN_ATTRS_DONE(attr_set);
  }
  bool has_res = false;   // one residual input anywhere -> the variant that carries the residual path
  for (int i = 0; i < a.nseg; ++i) has_res = has_res || a.seg[i].residual != nullptr;
  const dim3 g3(grid), b3(512);
  if (a.seg[0].bn_y) {   // data gradient + stage 1 of the BatchNorm backward reduction (validated by the caller)
    hipLaunchKernelGGL((conv_halo_kernel<false, false, true, SPLIT, WM>), g3, b3, LDS, st, a);
  } else if (out_f32) {
    if (has_res) hipLaunchKernelGGL((conv_halo_kernel<true, true, false, SPLIT, WM>), g3, b3, LDS, st, a);
    else hipLaunchKernelGGL((conv_halo_kernel<true, false, false, SPLIT, WM>), g3, b3, LDS, st, a);
  } else {
    if (has_res) hipLaunchKernelGGL((conv_halo_kernel<false, true, false, SPLIT, WM>), g3, b3, LDS, st, a);
    else hipLaunchKernelGGL((conv_halo_kernel<false, false, false, SPLIT, WM>), g3, b3, LDS, st, a);
  }
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// Whole tiles: one persistent workgroup per CU (minus the CUs kept for RCCL; opts.max_workgroups caps it).  With a
// split plan (rn_splitk_plan: a.split_s > 1) the launch is TWO kernels: the full rounds — tiles [0, split_f), the
// round-3 kernel untouched — and the SPLIT instantiation for the tiles of the last round, one part per workgroup.  (The
// unit mapping inside the persistent loop cost the whole-tile kernel 10 % — registers — for a 5 % shorter tail.)
// wm = 4: the 512 x 128 tiles of the narrow layers (whole tiles only).
int rn_launch_conv_halo(const ConvArgs& a, bool out_f32, const rn_launch_opts& opts, hipStream_t st, int wm) {
  if (wm == 4) return halo_launch<false, 4>(a, out_f32, rn_persistent_grid(a.total_tiles, rn_num_cus(), opts), st);
  if (a.split_s <= 1) return halo_launch<false, 2>(a, out_f32, rn_persistent_grid(a.total_tiles, rn_num_cus(), opts), st);
  if (a.split_f > 0) {
    ConvArgs full = a;
    full.total_tiles = a.split_f;   // tile_of() then numbers the first split_f tiles
    const int rc = halo_launch<false, 2>(full, out_f32, rn_persistent_grid(full.total_tiles, rn_num_cus(), opts), st);
    if (rc != RN_OK) return rc;
  }
  return halo_launch<true, 2>(a, out_f32, a.vtotal, st);
}

// patch pixels the worst tile of an [N, H, W] tensor needs (rows from one above its first pixel row to one below
// its last in the shared-pad numbering, W + 1 apart, + the closing zero pixel); the caller compares it with the
// kernel's capacity
// Patch pixels per image row: W + 1 (the shared zero column).  A pitch rounded up to 8 pixels (one 128-byte bank row per
// plane, so that a fragment crossing a row seam stays conflict free: 17 % of the LDS cycles are bank conflicts at the
// seams, profiles/r01_pmc_train_b32_lds_conflicts.csv) was measured 0.6 % SLOWER on the training step (34.87 / 34.96 ms
// -> 35.11 / 35.13 ms: the 9 % more patch bytes through the LDS-DMA cost more than the conflicts did) and was removed.
int rn_conv_halo_pitch(int W) { return W + 1; }

int rn_conv_halo_patch_pixels(int N, int H, int W, int pitch, int BM) {   // BM: pixels per tile, 256 or 512
  const long long M = (long long)N * H * W, HW = (long long)H * W;
  long long worst = 0;
  for (long long m0 = 0; m0 < M; m0 += BM) {
    const long long ml = m0 + BM - 1 < M ? m0 + BM - 1 : M - 1;
    const long long nf = m0 / HW, nl = ml / HW;
    const long long Gf = nf * (H + 1) + (m0 - nf * HW) / W + 1, Gl = nl * (H + 1) + (ml - nl * HW) / W + 1;
    worst = worst > Gl - Gf + 3 ? worst : Gl - Gf + 3;
  }
  return (int)(worst * pitch + 1);
}
int rn_conv_halo_capacity(int BM) { return BM == 512 ? HaloGeo<4>::PIX_PX : HaloGeo<2>::PIX_PX; }
