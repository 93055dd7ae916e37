// rn_wgrad_big.hip — weight gradient with 256 (co) x 256 (ci) per-tap tiles for the large layers
// (Cout >= 256 and Cin >= 256: head towers, class prediction, FPN 3x3, ResNet stage 3/4).
//
// Same GEMM as rn_wgrad.hip (dW[co][tap][ci] = sum_p dy[p][co] * x[p + tap][ci], K = pixels, both operands
// pixel-major in HBM, fragments via the transpose read ds_read_b64_tr_b16), rebuilt on the scheme that
// doubled the forward kernel (rn_conv_big.hip): the 128 x 128 tile issues 8 LDS-DMA instructions (~100 cycles
// each) per 16 MFMAs (512 cycles) per wave and sits at ~470 TFLOP/s on the head convs; here
//   * 512 threads = 8 waves as 2 (co) x 4 (ci), wave tile 128 x 64 = 4 x 2 MFMA 32x32x16 tiles, K step 32
//     pixels, four LDS stages of 32 KB ([32 px][256 ch] bf16 for dy and for x), 4 DMA pieces per wave per
//     16 MFMAs;
//   * the two waves of a SIMD ping-pong exactly as in conv_big_kernel: compute segment = 16 MFMAs from
//     registers, load segment = 24 transpose reads of the next step + this wave's 4 DMA pieces of the step
//     3-4 ahead + counted vmcnt(8); two s_barriers per K step;
//   * per DMA row the (image, y, x) coordinates advance incrementally; x rows of a shifted tap that fall
//     outside the image, rows past the pixel chunk and channel chunks past the tensor width get an
//     out-of-range buffer offset (hardware zero fill).
// One workgroup = one (co tile, ci tile, tap, pixel chunk); partial tiles go to the same workspace layout as
// rn_wgrad.hip and are summed by its deterministic reduce kernel.
#include <algorithm>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "rn_wgrad_dev.h"

namespace {

constexpr int BK = 32, STAGES = 4;
constexpr int OP_BYTES = BK * 512;            // one operand tile: 32 pixel rows x 256 channels bf16
constexpr int STAGE_BYTES = 2 * OP_BYTES;
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

// Transpose-read fragments from a [pixel][256 ch] tile (512-byte rows, 16-byte slot ^ ((row & 3) << 2)).
// Lane (g = lane>>4, i = lane&15) of a 32-channel MFMA tile starting at channel col0 reads the 4-pixel column
// blocks at rows (g>>1)*8 + (i>>2) [+4] (+16 for the second K slice), channels col0 + 16*(g&1) + 4*(i&3):
// the swizzle term only depends on the lane, so one byte offset per tile plus immediates covers all reads.
__device__ __forceinline__ int frag_off(int lane, int col0) {
  const int g = lane >> 4, i = lane & 15;
  const int col = col0 + 16 * (g & 1) + 4 * (i & 3);
  const int r0 = (g >> 1) * 8 + (i >> 2);
  return r0 * 512 + (((col >> 3) ^ ((r0 & 3) << 2)) << 4) + (col & 7) * 2;
}

// LINEAR: stride 1, output size == input size, symmetric padding (every 1x1 / 3x3 layer of the network): the input
// pixel of output pixel p at tap (r, s) is p + (r - pt)*W + (s - pl), so both operands' source offsets advance by a
// constant per K step and only (ox, oy) are tracked, for the border test of the shifted taps — 17 VALU per DMA row
// pair instead of ~45 (the load segment of this kernel is instruction-issue bound: ~230 instructions per K step
// against 16 MFMAs).
template <bool LINEAR>
__global__ void __launch_bounds__(512) wgrad_big_kernel(const WgArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (chunk, tile) from the block id: tiles of one pixel chunk are consecutive on one XCD (they walk the
  // chunk in lockstep and share its rows in L2)
  const int tiles_per_tap = args.co_tiles * args.ci_tiles;
  const int tiles_all = tiles_per_tap * args.R * args.S;
  // One work item per workgroup when the grid covers them all (the single-GPU case); with CUs kept free for RCCL
  // (rn_launch_opts.reserved_cus) the grid is smaller than the item count and a workgroup walks items bid, bid + grid, ...
#pragma unroll 1
  for (int bid = blockIdx.x; bid < tiles_all * args.total_chunks; bid += gridDim.x) {
  int logical;
  {
    const int total = tiles_all * args.total_chunks;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, rr = total & 7;
    logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  }
  const int chunk = logical / tiles_all;
  const int tile_id = logical - chunk * tiles_all;
  const int tap = tile_id / tiles_per_tap;
  const int tt = tile_id - tap * tiles_per_tap;
  const int co_t = tt / args.ci_tiles, ci_t = tt - co_t * args.ci_tiles;
  const int co0 = co_t * 256, ci0 = ci_t * 256;
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
  const int wave_m = wave >> 2, wave_n = wave & 3;   // wave_m is also the ping-pong group
  const unsigned lds0 = rn_lds_addr(smem);

  const __amdgpu_buffer_rsrc_t rs_dy =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.dy, 0, (int)((long long)sg.P * sg.dyS * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.x, 0, (int)((long long)sg.N * H * W * sg.xS * 2), 0x00020000);

  // DMA bookkeeping: piece q of an operand tile = pixel rows 2q, 2q+1 (lanes 0-31 / 32-63, one 16-byte
  // slot each); this wave issues pieces wave and wave + 8 of both operands
  const int d_row = lane >> 5, d_pos = lane & 31;
  // rows 2*wave + d_row and that + 16: the same (row & 3), hence the same channel chunk for both
  const int row_lo = 2 * wave + d_row;
  unsigned a_ch, b_ch;             // element offset of this lane's channel chunk, or OOB
  int r_ox[2], r_oy[2], r_n[2];    // (n, oy, ox) of this lane's two rows at the next step to issue
  {
    const int chunkc = d_pos ^ ((row_lo & 3) << 2);
    a_ch = (co0 + chunkc * 8 < sg.dyS) ? (unsigned)(co0 + chunkc * 8) : WG_OOB;
    b_ch = (ci0 + chunkc * 8 < Cin) ? (unsigned)(ci0 + chunkc * 8) : WG_OOB;
    const float inv_wo = 1.0f / (float)Wo, inv_ho = 1.0f / (float)Ho;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = row_lo + 16 * j;
      const int p0 = p_begin + row;
      int t2 = (int)((float)p0 * inv_wo);
      int ox = p0 - t2 * Wo;
      if (ox < 0) { ox += Wo; --t2; } else if (ox >= Wo) { ox -= Wo; ++t2; }
      int n = (int)((float)t2 * inv_ho);
      int oy = t2 - n * Ho;
      if (oy < 0) { oy += Ho; --n; } else if (oy >= Ho) { oy -= Ho; ++n; }
      r_ox[j] = ox; r_oy[j] = oy; r_n[j] = n;
    }
  }
  const int adv_q = BK / Wo, adv_r = BK - adv_q * Wo;            // 32 pixels = adv_q rows + adv_r columns
  const int adv_qn = adv_q / Ho, adv_qr = adv_q - adv_qn * Ho;   //           = adv_qn images + adv_qr rows + ...

  const int ksteps = (p_end - p_begin + BK - 1) / BK;
  int g_iss = 0;   // K steps issued by this wave
  // LINEAR state: byte offsets of this lane's two rows in dy and x (out of range for good where the channel chunk
  // is), the taps' forbidden output row / column (-1: none)
  unsigned l_dy[2], l_x[2];
  const int y_bad = r < args.pt ? 0 : (r > args.pt ? Ho - 1 : -1), x_bad = s < args.pl ? 0 : (s > args.pl ? Wo - 1 : -1);
  if (LINEAR) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p0 = p_begin + row_lo + 16 * j;
      l_dy[j] = a_ch == WG_OOB ? WG_OOB : (unsigned)((p0 * sg.dyS + (int)a_ch) * 2);
      // the shifted pixel of a valid tap is inside the same image; invalid ones are masked at issue
      l_x[j] = b_ch == WG_OOB ? WG_OOB : (unsigned)(((p0 + (r - args.pt) * W + (s - args.pl)) * sg.xS + (int)b_ch) * 2);
    }
  }
  const unsigned l_dy_step = (unsigned)(BK * sg.dyS * 2), l_x_step = (unsigned)(BK * sg.xS * 2);

#define WGB_ISSUE()                                                                               \
  do {                                                                                            \
    char* st__ = smem + (g_iss & 3) * STAGE_BYTES;                                                \
    const int pb__ = p_begin + g_iss * BK;                                                        \
    if (LINEAR) {                                                                                 \
      const int left__ = p_end - pb__;   /* rows of this step that are inside the chunk */         \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                             \
        const bool in__ = row_lo + 16 * j < left__;                                               \
        dma16(rs_dy, st__ + (wave + 8 * j) * 1024, in__ ? l_dy[j] : WG_OOB);                      \
        const bool ok__ = in__ && r_oy[j] != y_bad && r_ox[j] != x_bad;                           \
        dma16(rs_x, st__ + OP_BYTES + (wave + 8 * j) * 1024, ok__ ? l_x[j] : WG_OOB);             \
        l_dy[j] += l_dy_step;                                                                     \
        l_x[j] += l_x_step;                                                                       \
        int ox__ = r_ox[j] + adv_r, oy__ = r_oy[j] + adv_qr;                                      \
        const int c1__ = ox__ >= Wo;                                                              \
        ox__ -= c1__ ? Wo : 0; oy__ += c1__;                                                      \
        oy__ -= oy__ >= Ho ? Ho : 0;                                                              \
        r_ox[j] = ox__; r_oy[j] = oy__;                                                           \
      }                                                                                           \
    } else {                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
      const int p__ = pb__ + row_lo + 16 * j;                                                     \
      const bool in__ = p__ < p_end;                                                              \
      const unsigned va__ = (in__ && a_ch != WG_OOB)                                              \
                                ? (__umul24((unsigned)p__, (unsigned)sg.dyS) + a_ch) * 2u : WG_OOB; \
      dma16(rs_dy, st__ + (wave + 8 * j) * 1024, va__);                                           \
      const int iy__ = r_oy[j] * args.sh - args.pt + r, ix__ = r_ox[j] * args.sw - args.pl + s;   \
      const bool ok__ = in__ && b_ch != WG_OOB && (unsigned)iy__ < (unsigned)H && (unsigned)ix__ < (unsigned)W; \
      const unsigned pix__ = __umul24(__umul24((unsigned)r_n[j], (unsigned)H) + (unsigned)iy__, (unsigned)W) + (unsigned)ix__; \
      const unsigned vb__ = ok__ ? (__umul24(pix__, (unsigned)sg.xS) + b_ch) * 2u : WG_OOB;       \
      dma16(rs_x, st__ + OP_BYTES + (wave + 8 * j) * 1024, vb__);                                 \
      int ox__ = r_ox[j] + adv_r, oy__ = r_oy[j] + adv_qr, n__ = r_n[j] + adv_qn;                 \
      const int c1__ = ox__ >= Wo;                                                                \
      ox__ -= c1__ ? Wo : 0; oy__ += c1__;                                                        \
      const int c2__ = oy__ >= Ho;                                                                \
      oy__ -= c2__ ? Ho : 0; n__ += c2__;                                                         \
      const int c3__ = oy__ >= Ho;                                                                \
      oy__ -= c3__ ? Ho : 0; n__ += c3__;                                                         \
      r_ox[j] = ox__; r_oy[j] = oy__; r_n[j] = n__;                                               \
    }                                                                                             \
    }                                                                                             \
    ++g_iss;                                                                                      \
  } while (0)

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

  bf16x8_t fa0[4], fb0[2], fa1[4], fb1[2];   // fragments of one K step: pixels 0..15 / 16..31
  int offa[4], offb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) offa[i] = frag_off(lane, wave_m * 128 + i * 32);
#pragma unroll
  for (int j = 0; j < 2; ++j) offb[j] = OP_BYTES + frag_off(lane, wave_n * 64 + j * 32);
// 24 transpose reads from inline asm (rn_wgrad_dev.h: through the builtin the compiler drains the whole DMA ring — an
// `s_waitcnt vmcnt(0)` in front of the first read of every K step — and the counted vmcnt(8) below is moot); they return
// while this wave issues its DMA pieces; ONE wait names every destination
#define WGB_READ(stage_)                                                                          \
  do {                                                                                            \
    const unsigned t__ = lds0 + (unsigned)((stage_) * STAGE_BYTES);                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
      const unsigned a__ = t__ + (unsigned)offb[j];                                               \
      RN_TR_ISSUE(q[4 * j], a__, 0); RN_TR_ISSUE(q[4 * j + 1], a__, 2048);                        \
      RN_TR_ISSUE(q[4 * j + 2], a__, 8192); RN_TR_ISSUE(q[4 * j + 3], a__, 8192 + 2048);          \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      const unsigned a__ = t__ + (unsigned)offa[i];                                               \
      RN_TR_ISSUE(q[8 + 4 * i], a__, 0); RN_TR_ISSUE(q[8 + 4 * i + 1], a__, 2048);                \
      RN_TR_ISSUE(q[8 + 4 * i + 2], a__, 8192); RN_TR_ISSUE(q[8 + 4 * i + 3], a__, 8192 + 2048);  \
    }                                                                                             \
  } while (0)
#define WGB_LOADSEG(stage_)                                                                       \
  do {                                                                                            \
    rn_u32x2_t q[24];                                                                             \
    WGB_READ(stage_);                                                                             \
    if (g_iss < ksteps) {                                                                         \
      WGB_ISSUE();                                                                                \
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                            \
    } else {                                                                                      \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
    }                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                           \
                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]),         \
                   "+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]),   \
                   "+v"(q[16]), "+v"(q[17]), "+v"(q[18]), "+v"(q[19]), "+v"(q[20]), "+v"(q[21]), "+v"(q[22]), "+v"(q[23])  \
                 :: "memory");                                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
      fb0[j] = rn_tr_frag(q[4 * j], q[4 * j + 1]); fb1[j] = rn_tr_frag(q[4 * j + 2], q[4 * j + 3]); \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      fa0[i] = rn_tr_frag(q[8 + 4 * i], q[8 + 4 * i + 1]); fa1[i] = rn_tr_frag(q[8 + 4 * i + 2], q[8 + 4 * i + 3]); \
    }                                                                                             \
  } while (0)
#define WGB_COMPUTESEG()                                                                          \
  do {                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
        acc[i][j] = RN_MFMA_32x32x16(fa0[i], fb0[j], acc[i][j], 0, 0, 0);  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                               \
        acc[i][j] = RN_MFMA_32x32x16(fa1[i], fb1[j], acc[i][j], 0, 0, 0);  \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
#define WGB_BARRIER()                       \
  do {                                      \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

  // prologue: steps 0..2 in flight, 0 and 1 complete; pre-roll: both groups read step 0 and issue step 3
  // (group 1 as its slot-0 load segment).  Both groups execute 2 * ksteps barriers.
#pragma unroll 1
  for (int t = 0; t < 3 && g_iss < ksteps; ++t) WGB_ISSUE();
  if (g_iss >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WGB_BARRIER();
  int g = 0;
  // one loop for both groups: group 1 runs one barrier ahead, group 0 closes with the matching one
  WGB_LOADSEG(0);
  if (wave_m == 1) WGB_BARRIER();
#pragma unroll 1
  while (true) {
    WGB_COMPUTESEG();
    WGB_BARRIER();
    if (++g == ksteps) break;
    WGB_LOADSEG(g & 3);
    WGB_BARRIER();
  }
  if (wave_m == 0) WGB_BARRIER();
#undef WGB_BARRIER
#undef WGB_COMPUTESEG
#undef WGB_LOADSEG
#undef WGB_READ
#undef WGB_ISSUE

  // partial tile -> workspace[chunk][co][tap][ci]  (lane = ci column: 128-byte rows)
  // a fresh lane id so that nothing of the epilogue is live across the loop (which runs at the register limit)
  const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int taps = args.R * args.S;
  const int row_stride = taps * Cin;                      // floats between consecutive co rows
  float* out = args.ws + (long long)chunk * Cout * row_stride + (long long)tap * Cin;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wave_n * 64 + j * 32 + (elane & 31);
      const int cob = co0 + wave_m * 128 + i * 32 + 4 * (elane >> 5);
      float* po = out + (long long)cob * row_stride + ci;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int dco = (q & 3) + 8 * (q >> 2);
        if (cob + dco < Cout && ci < Cin) po[(long long)dco * row_stride] = acc[i][j][q];
      }
    }
  __syncthreads();   // the next item's DMA reuses the stages
  }   // work items of this workgroup
}

}  // namespace


// Layers worth the 256 x 256 tile: both channel counts >= 256 and enough pixels to split K over the CUs.
bool rn_wgrad_big_plan(const rn_wgrad_problem* p, WgArgs& a) {
  if (a.Cin < 256 || a.Cout < 256) return false;
  long long Ptot = 0;
  for (int i = 0; i < p->num_segments; ++i) Ptot += (long long)p->seg[i].N * p->seg[i].Ho * p->seg[i].Wo;
  const int co_tiles = (int)rn_cdiv(a.Cout, 256), ci_tiles = (int)rn_cdiv(a.Cin, 256);
  const int tiles = co_tiles * ci_tiles * a.R * a.S;
  if ((Ptot < 16384 && p->opts.wgrad_kernel < 2) || tiles > 512) return false;
  a.co_tiles = co_tiles;
  a.ci_tiles = ci_tiles;
  a.co_groups = 1;
  a.gco = co_tiles;
  // One workgroup per CU (128 KB of LDS).  Candidates: chunk lengths that give about 1, 1.5, 2 and 3 rounds of
  // the 256 CUs; segments (pyramid levels) are chunked separately, so each candidate grows its chunk until the
  // workgroup count fits.  The kernel hands XCD x a contiguous range of (chunk, tile) ids, and a launch whose
  // tile count does not divide the 32 CUs of an XCD (27 tiles for 256 -> 720) leaves one round badly filled:
  // the candidates are priced with a greedy simulation of that mapping (per-workgroup cost = K steps + a
  // fixed prologue / epilogue / partial-tile cost) and the cheapest wins.  Plans are cached per shape.
  long long CH = 0;
  // one round of the CUs the kernel may use: fewest split-K partials (opts.wgrad_target_blocks: no candidate search)
  const bool g_big_target_user = p->opts.wgrad_target_blocks > 0;
  const long long g_big_target_blocks = g_big_target_user ? p->opts.wgrad_target_blocks : rn_num_cus() - p->opts.reserved_cus;
  {
    static std::mutex mu;
    static std::unordered_map<std::string, long long> cache;
    std::string key((const char*)&g_big_target_blocks, sizeof(g_big_target_blocks));
    key.push_back(g_big_target_user ? 1 : 0);
    const int dims[4] = {a.R * 16 + a.S, a.Cin, a.Cout, p->num_segments};
    key.append((const char*)dims, sizeof(dims));
    for (int i = 0; i < p->num_segments; ++i) key.append((const char*)&a.seg[i].P, sizeof(a.seg[i].P));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      CH = it->second;
    } else {
      const int mult_x2[4] = {2, 3, 4, 6};
      const int ncand = g_big_target_user ? 1 : 4;
      double best = 0;
      for (int c = 0; c < ncand; ++c) {
        const long long blocks = (long long)g_big_target_blocks * mult_x2[c] / 2;
        long long target = blocks / tiles;
        if (target < 1) target = 1;
        long long ch = rn_cdiv(rn_cdiv(Ptot, target), BK) * BK;
        if (ch < 4 * BK) ch = 4 * BK;
        int chunks = 0;
        for (int it2 = 0; it2 < 16; ++it2) {
          chunks = 0;
          for (int i = 0; i < p->num_segments; ++i) chunks += (int)rn_cdiv(a.seg[i].P, ch);
          if ((long long)chunks * tiles <= blocks || chunks <= 1) break;
          ch += BK * rn_cdiv(ch / BK, 16);    // +6 % per iteration
        }
        // greedy schedule of the kernel's XCD mapping: workgroup `logical` = chunk * tiles + tile
        const long long total = (long long)chunks * tiles;
        std::vector<int> steps;   // K steps of every chunk
        for (int i = 0; i < p->num_segments; ++i)
          for (long long b = 0; b < a.seg[i].P; b += ch)
            steps.push_back((int)rn_cdiv(std::min<long long>(ch, a.seg[i].P - b), BK));
        const double fixed = 30.0;   // prologue + epilogue + partial tile write, in K steps
        double makespan = 0;
        const long long q = total >> 3, rr = total & 7;
        long long begin = 0;
        for (int x = 0; x < 8; ++x) {
          const long long n = q + (x < rr ? 1 : 0);
          double cu[32];
          for (int k = 0; k < 32; ++k) cu[k] = 0;
          for (long long l = begin; l < begin + n; ++l) {
            int k0 = 0;
            for (int k = 1; k < 32; ++k) if (cu[k] < cu[k0]) k0 = k;
            cu[k0] += steps[(size_t)(l / tiles)] + fixed;
          }
          for (int k = 0; k < 32; ++k) makespan = std::max(makespan, cu[k]);
          begin += n;
        }
        if (c == 0 || makespan < best * 0.97) {   // more workgroups only for a clear gain
          best = makespan;
          CH = ch;
        }
      }
      cache.emplace(key, CH);
    }
  }
  a.CH = (int)CH;
  int chunks = 0;
  for (int i = 0; i < p->num_segments; ++i) {
    WgSegDev& d = a.seg[i];
    d.chunk_begin = chunks;
    chunks += (int)rn_cdiv(d.P, CH);
  }
  a.total_chunks = chunks;
  return true;
}

int rn_launch_wgrad_big(const WgArgs& a, const rn_launch_opts& opts, hipStream_t st) {
  static unsigned long long attr_set = 0;   // one bit per device
  if (RN_ATTRS_NEEDED(attr_set)) {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_big_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     LDS_BYTES));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_big_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     LDS_BYTES));
    RN_ATTRS_DONE(attr_set);
  }
  bool linear = a.sh == 1 && a.sw == 1 && a.pt == (a.R - 1) / 2 && a.pl == (a.S - 1) / 2 && (a.R & 1) && (a.S & 1);
  for (int i = 0; i < a.nseg; ++i) {
    const WgSegDev& s = a.seg[i];
    linear = linear && s.Ho == s.H && s.Wo == s.W &&
             (long long)s.N * s.H * s.W * (s.xS > s.dyS ? s.xS : s.dyS) * 2 < (1ll << 31) - (1ll << 24);
  }
  int items = a.co_tiles * a.ci_tiles * a.R * a.S * a.total_chunks;
  rn_launch_opts o = opts;
  o.max_workgroups = 0;   // the cap is for the persistent convolution grids
  dim3 grid((unsigned)(opts.reserved_cus > 0 ? rn_persistent_grid(items, rn_num_cus(), o) : items));
  if (linear) hipLaunchKernelGGL(wgrad_big_kernel<true>, grid, dim3(512), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(wgrad_big_kernel<false>, grid, dim3(512), LDS_BYTES, st, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
