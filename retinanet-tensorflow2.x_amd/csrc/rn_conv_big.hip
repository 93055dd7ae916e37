// rn_conv_big.hip — 256 x 256 x 32 implicit-GEMM conv tile for the MFMA-bound layers (Cout >= 256).
//
// Why a second shape: the 128 x 128 tile of rn_conv.hip needs (128+128)*2 B of staged operands per
// 2*128*128 FLOP = 64 B/clk/CU at the MFMA peak, i.e. 100 % of the vector-memory -> LDS path; measured
// it sustains ~22 B/clk/CU and 34 % of the MFMA peak.  A 256 x 256 tile halves the
// bytes per FLOP (32 B/clk/CU at peak) and halves the LDS-DMA instructions per MFMA.
//
//   * 512 threads = 8 wavefronts as 2 (M) x 4 (N), wave tile 128 pixels x 64 channels = 4 x 2 MFMA
//     32x32x16 tiles, 128 accumulator registers; two waves per SIMD interleave MFMA with the other
//     wave's ds_read / DMA issue.
//   * K step 32, four LDS stages of 32 KB (128 KB, one workgroup per CU): `buffer_load ... lds` DMA runs
//     three K steps ahead behind a counted `s_waitcnt vmcnt(4)`, one raw s_barrier per K step.
//   * fragments are software pipelined by half K step (two register sets of 6 x ds_read_b128): the
//     reads of the next 16-wide K slice are in flight while the 8 MFMAs of the current one run.
//   * operands are swapped (weights = MFMA A, pixels = MFMA B), so a lane's accumulator registers are
//     4 consecutive output channels of ONE pixel: the epilogue applies scale/shift, rounds to bf16 and
//     transposes through a per-wave 32 x 64 LDS patch with 8-byte writes, then reads 16 bytes
//     (8 channels) per lane and stores full 128-byte rows; residual add + activation happen on the
//     read-back side (the conv+BN value is rounded to bf16 before the add, as the reference's bf16
//     BatchNormalization output is).  No workgroup barrier in the epilogue.
#include "rn_conv_dev.h"

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int BM = 256, BN = 256, BK = 32, NW = 8, STAGES = 4;
constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;


__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);  // v_cvt_pk_bf16_f32: RNE, like rn_f32_to_bf16
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ABL (tools/bench_conv.py --ablate, 0 in production): 1 = no DMA inside the K loop, 2 = no ds_read inside
// the K loop, 4 = no MFMA, 8 = no epilogue.
__device__ long long g_big_timing[8 * 8];   // [wave][phase] cycle sums of workgroup 300 (ABL & 64)

template <bool OUT_F32, int ABL = 0>
__global__ void __launch_bounds__(512) conv_big_kernel(const ConvArgs args) {
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_prev = 0;
#define BIG_STAMP(k_)                              \
  do {                                             \
    if (ABL & 64) {                                \
      const long long now__ = clock64();           \
      tm[k_] += now__ - t_prev;                    \
      t_prev = now__;                              \
    }                                              \
  } while (0)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- tile lookup (XCD-aware remap of the linear block id) ---------------------------------
  int tile;
  {
    const int total = args.total_tiles;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = total >> 3, r = total & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
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
  const int wave_m = wave >> 2, wave_n = wave & 3;

  const int S = args.S;
  const int H = sg.H, W = sg.W, Cin = sg.CinP, PS = sg.pix_stride;
  const int M = sg.M;
  const int Ktot = args.R * S * Cin;
  const int cout_rows = sg.Cout <= 64 ? 64 : ((sg.Cout + 127) / 128) * 128;  // rows the packed weights hold

  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.x, 0, (int)((long long)sg.N * H * W * PS * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)sg.w, 0, (int)((long long)cout_rows * Ktot * 2), 0x00020000);

  // ---- per-lane DMA bookkeeping: instruction j of this wave fills rows (j*8 + wave)*16 + lane/4 ----
  const int d_row = lane >> 2, d_pos = lane & 3;
  unsigned a_off[2], a_mask[2], b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (j * NW + wave) * 16 + d_row;
    const int chunk = d_pos ^ lds_swz<BK>(row);
    const int m = m0 + row;
    const int mm = m < M ? m : 0;
    const int ox = mm % sg.Wo;
    const int t2 = mm / sg.Wo;
    const int oy = t2 % sg.Ho;
    const int n = t2 / sg.Ho;
    const int iy0 = oy * args.sh - args.pt, ix0 = ox * args.sw - args.pl;
    a_off[j] = (unsigned)(((((long long)n * H + iy0) * W + ix0) * PS + chunk * 8) * 2);
    unsigned mask = 0;
    if (m < M) {
      for (int r = 0; r < args.R; ++r)
        for (int s = 0; s < S; ++s) {
          const bool ok = (unsigned)(iy0 + r) < (unsigned)H && (unsigned)(ix0 + s) < (unsigned)W;
          mask |= (ok ? 1u : 0u) << (r * S + s);
        }
    }
    a_mask[j] = mask;
    const int nrow = n0 + row;
    b_off[j] = nrow < cout_rows ? (unsigned)(((long long)nrow * Ktot + chunk * 8) * 2) : RN_OOB;
  }

  // ---- fragment read offsets ---------------------------------------------------------------------
  // row = lane&31 of a 32-row tile, 16-byte slot = 2*kk + (lane>>5), XOR-swizzled by (row/4)&3; the
  // swizzle only depends on lane&31 because every tile starts at a multiple of 16 rows.
  const int fr = lane & 31, fh = lane >> 5;
  const int sw = (fr >> 2) & 3;
  const int off_p0 = (wave_m * 128 + fr) * 64 + ((0 + fh) ^ sw) * 16;            // pixels, kk = 0
  const int off_p1 = (wave_m * 128 + fr) * 64 + ((2 + fh) ^ sw) * 16;            // pixels, kk = 1
  const int off_w0 = A_BYTES + (wave_n * 64 + fr) * 64 + ((0 + fh) ^ sw) * 16;   // weights
  const int off_w1 = A_BYTES + (wave_n * 64 + fr) * 64 + ((2 + fh) ^ sw) * 16;

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps = args.R * S * (Cin / BK);

#define BIG_ISSUE(stage_, tap_, c0_)                                                          \
  do {                                                                                        \
    const int r__ = (tap_) / S, s__ = (tap_) - r__ * S;                                       \
    const unsigned tap_off__ = (unsigned)((((long long)r__ * W + s__) * PS + (c0_)) * 2);     \
    const unsigned koff__ = (unsigned)(((long long)(tap_) * Cin + (c0_)) * 2);                \
    char* st__ = smem + (stage_) * STAGE_BYTES;                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                           \
      unsigned v__ = ((a_mask[j] >> (tap_)) & 1u) ? a_off[j] + tap_off__ : RN_OOB;            \
      if (ABL & 128) v__ = (unsigned)(m0 * 64 + (j * NW + wave) * 1024 + lane * 16) + tap_off__; \
      dma16(rs_x, st__ + (j * NW + wave) * 1024, v__);                                        \
    }                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                           \
      unsigned v__ = b_off[j] == RN_OOB ? RN_OOB : b_off[j] + koff__;                         \
      if (ABL & 128) v__ = (unsigned)((j * NW + wave) * 1024 + lane * 16) + koff__;           \
      dma16(rs_w, st__ + A_BYTES + (j * NW + wave) * 1024, v__);                              \
    }                                                                                         \
  } while (0)
#define BIG_ADVANCE()  \
  do {                 \
    c0 += BK;          \
    if (c0 >= Cin) {   \
      c0 = 0;          \
      ++tap;           \
    }                  \
  } while (0)
#define BIG_LOAD(PX, WT, stage_, offp_, offw_)                                                \
  do {                                                                                        \
    const char* b__ = smem + (stage_) * STAGE_BYTES;                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) PX[i] = *(const bf16x8_t*)(b__ + (offp_) + i * 2048); \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) WT[j] = *(const bf16x8_t*)(b__ + (offw_) + j * 2048); \
  } while (0)
#define BIG_MFMA(PX, WT)                                                                      \
  do {                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WT[j], PX[i], acc[i][j], 0, 0, 0); \
  } while (0)

  int tap = 0, c0 = 0;  // coordinates of the NEXT tile to issue
  // group 0 (which loads in the odd slots) runs one tile further ahead than group 1, so both get
  // four slots (two K steps) between issuing a piece and having to publish it
  const int pre = wave_m == 0 ? (ksteps < 4 ? ksteps : 4) : (ksteps < 3 ? ksteps : 3);
#pragma unroll 1
  for (int t = 0; t < pre; ++t) {
    BIG_ISSUE(t, tap, c0);
    BIG_ADVANCE();
  }
  int issued = pre;
  {
    // group 0 needs its pieces of tiles 0 and 1 landed, group 1 those of tile 0
    const int pending = wave_m == 0 ? pre - 2 : pre - 1;
    if (pending >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (pending == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- ping-pong main loop ---------------------------------------------------------------------------
  // The two waves of a SIMD (wave w and w+4: groups 0 and 1) alternate roles every half K step: one runs
  // its 16 MFMAs back to back from registers while the other waits for its DMA pieces, issues the pieces
  // of the tile three steps ahead and reads its 12 fragments of the next tile; two s_barriers per K step.
  //   slot 2t  : group 0 MFMA(t)                   | group 1 LOAD(read tile t, issue tile t+3)
  //   slot 2t+1: group 0 LOAD(read t+1, issue t+4) | group 1 MFMA(t)
  // Tile T is read first in slot 2T-1 (group 0) and its stage was last read (tile T-4, group 1) in slot
  // 2T-8, so group 1 issues it in slot 2T-6 and group 0 in slot 2T-7.  At the END of a load segment a
  // wave waits for everything but the pieces of this and the previous segment (vmcnt 8) and the barrier
  // that ends the segment publishes them: group 1's pieces of tile T in slot 2T-2, group 0's in 2T-3.
  bf16x8_t px0[4], wt0[2], px1[4], wt1[2];
#define BIG_LOADSEG(read_stage_)                                                              \
  do {                                                                                        \
    BIG_STAMP(7);                                                                             \
    /* fragment reads first: their LDS latency runs under the DMA issue below (the reverse    \
       order on half of the waves measured 6 % slower) */                                     \
    if (!(ABL & 2)) {                                                                         \
      BIG_LOAD(px0, wt0, read_stage_, off_p0, off_w0);                                        \
      BIG_LOAD(px1, wt1, read_stage_, off_p1, off_w1);                                        \
    }                                                                                         \
    BIG_STAMP(0);                                                                             \
    const bool do_issue__ = issued < ksteps && !(ABL & 1);                                    \
    if (do_issue__) {                                                                         \
      BIG_ISSUE(issued & 3, tap, c0);                                                         \
      BIG_ADVANCE();                                                                          \
    }                                                                                         \
    BIG_STAMP(1);                                                                             \
    /* everything but the pieces of this and the previous load segment has landed: the        \
       barrier that ends the segment publishes it */                                          \
    if (do_issue__) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                          \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
    BIG_STAMP(2);                                                                             \
    ++issued;                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
    BIG_STAMP(3);                                                                             \
  } while (0)
#define BIG_MFMASEG()                                                                         \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (!(ABL & 4)) {                                                                         \
      BIG_MFMA(px0, wt0);                                                                     \
      BIG_MFMA(px1, wt1);                                                                     \
    } else {                                                                                  \
      acc[0][0][0] += (float)px0[0][0] + (float)wt0[0][0] + (float)px1[0][0] + (float)wt1[0][0] +       \
                      (float)px0[3][7] + (float)wt0[1][7] + (float)px1[3][7] + (float)wt1[1][7] +       \
                      (float)px0[1][3] + (float)px0[2][5] + (float)px1[1][3] + (float)px1[2][5];        \
    }                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#define BIG_BARRIER()                       \
  do {                                      \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

  if (ABL & 64) t_prev = clock64();
  if (wave_m == 0) {
    BIG_LOAD(px0, wt0, 0, off_p0, off_w0);
    BIG_LOAD(px1, wt1, 0, off_p1, off_w1);
#pragma unroll 1
    for (int kt = 0; kt < ksteps; ++kt) {
      BIG_MFMASEG();
      BIG_STAMP(5);
      BIG_BARRIER();
      BIG_STAMP(6);
      BIG_LOADSEG((kt + 1) & 3);   // the last one reads a stale stage into registers nobody uses
      BIG_BARRIER();
      BIG_STAMP(4);
    }
  } else {
#pragma unroll 1
    for (int kt = 0; kt < ksteps; ++kt) {
      BIG_LOADSEG(kt & 3);
      BIG_BARRIER();
      BIG_STAMP(4);
      BIG_MFMASEG();
      BIG_STAMP(5);
      BIG_BARRIER();
      BIG_STAMP(6);
    }
  }
#undef BIG_BARRIER
#undef BIG_MFMASEG
#undef BIG_LOADSEG
  if ((ABL & 64) && blockIdx.x == 300 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_big_timing[wave * 8 + k] = tm[k];
  }
#undef BIG_MFMA
#undef BIG_LOAD
#undef BIG_ADVANCE
#undef BIG_ISSUE

  // every wave is done with the staging ring before it is reused as per-wave transpose patches
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  if (ABL & 8) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) ((float*)sg.y)[tid] = t;
    return;
  }
  // ---- epilogue ------------------------------------------------------------------------------
  // acc[i][j][r]: pixel m = m0 + wave_m*128 + i*32 + (lane&31),
  //               channel n = n0 + wave_n*64 + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
  const int Cout = sg.Cout;
  const int nw0 = n0 + wave_n * 64;
  char* patch = smem + wave * 8192;  // 32 pixels x 64 channels, bf16 (4 KB) or f32 (8 KB)
  if (!OUT_F32) {
    // read-back role: 8 lanes per pixel row (16 B = 8 channels each), 8 rows per pass
    const int rrow = lane >> 3, ru = lane & 7;
    const int nr = nw0 + ru * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + g * 8 + fh * 4;   // channel inside the wave's 64
          const int n = nw0 + nl;
          float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n < Cout && !(ABL & 32)) {
            if (sg.scale) sc = *(const float4*)(sg.scale + n);
            if (sg.shift) sf = *(const float4*)(sg.shift + n);
          }
          uint2 pk;
          pk.x = pack2(acc[i][j][g * 4 + 0] * sc.x + sf.x, acc[i][j][g * 4 + 1] * sc.y + sf.y);
          pk.y = pack2(acc[i][j][g * 4 + 2] * sc.z + sf.z, acc[i][j][g * 4 + 3] * sc.w + sf.w);
          // 16-byte unit (nl/8) swizzled by the pixel row; the 8-byte half stays in place
          *(uint2*)(patch + fr * 128 + (((nl >> 3) ^ (fr & 7)) << 4) + (nl & 4) * 2) = pk;
        }
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + rrow;
        const uint4 v = *(const uint4*)(patch + row * 128 + ((ru ^ (row & 7)) << 4));
        const int m = m0 + wave_m * 128 + i * 32 + row;
        if ((ABL & 16) && v.x != 0x12345678u) continue;
        if (m < M && nr < Cout) {
          const long long o = (long long)m * Cout + nr;
          float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
          if (sg.residual) {
            const uint4 rv = *(const uint4*)(sg.residual + o);
            f[0] += bf_lo(rv.x); f[1] += bf_hi(rv.x); f[2] += bf_lo(rv.y); f[3] += bf_hi(rv.y);
            f[4] += bf_lo(rv.z); f[5] += bf_hi(rv.z); f[6] += bf_lo(rv.w); f[7] += bf_hi(rv.w);
          }
          uint4 ov;
          if (args.act == RN_ACT_NONE && !sg.residual) {
            ov = v;
          } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = rn_apply_act(f[q], args.act);
            ov.x = pack2(f[0], f[1]); ov.y = pack2(f[2], f[3]); ov.z = pack2(f[4], f[5]); ov.w = pack2(f[6], f[7]);
          }
          *(uint4*)((uint16_t*)sg.y + o) = ov;
        }
      }
    }
  } else {
    // f32 output (prediction convs): 16 lanes per pixel row (16 B = 4 channels each), 4 rows per pass
    const int rrow = lane >> 4, ru = lane & 15;
    const int nr = nw0 + ru * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + g * 8 + fh * 4;
          const int n = nw0 + nl;
          float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n < Cout) {
            if (sg.scale) sc = *(const float4*)(sg.scale + n);
            if (sg.shift) sf = *(const float4*)(sg.shift + n);
          }
          float4 v;
          v.x = acc[i][j][g * 4 + 0] * sc.x + sf.x;
          v.y = acc[i][j][g * 4 + 1] * sc.y + sf.y;
          v.z = acc[i][j][g * 4 + 2] * sc.z + sf.z;
          v.w = acc[i][j][g * 4 + 3] * sc.w + sf.w;
          *(float4*)(patch + fr * 256 + (((nl >> 2) ^ (fr & 15)) << 4)) = v;
        }
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 4 + rrow;
        float4 v = *(const float4*)(patch + row * 256 + ((ru ^ (row & 15)) << 4));
        const int m = m0 + wave_m * 128 + i * 32 + row;
        if (m < M && nr < Cout) {
          const long long o = (long long)m * Cout + nr;
          if (sg.residual) {
            const uint2 rv = *(const uint2*)(sg.residual + o);
            v.x += bf_lo(rv.x); v.y += bf_hi(rv.x); v.z += bf_lo(rv.y); v.w += bf_hi(rv.y);
          }
          v.x = rn_apply_act(v.x, args.act);
          v.y = rn_apply_act(v.y, args.act);
          v.z = rn_apply_act(v.z, args.act);
          v.w = rn_apply_act(v.w, args.act);
          *(float4*)((float*)sg.y + o) = v;
        }
      }
    }
  }
}

}  // namespace

// internal (tools/bench_conv.py): per-phase cycle sums recorded by the ABL=64 build
extern "C" int rn_debug_conv_big_timing(long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_big_timing), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}

static int g_big_ablate = 0;
extern "C" void rn_debug_conv_big_ablate(int mask) { g_big_ablate = mask; }

template <int ABL>
static int launch_big_ablate(const ConvArgs& a, hipStream_t st) {
  constexpr int lds = STAGES * STAGE_BYTES;
  RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<false, ABL>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((conv_big_kernel<false, ABL>), dim3(a.total_tiles), dim3(512), lds, st, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

int rn_launch_conv_big(const ConvArgs& a, bool out_f32, hipStream_t st) {
  constexpr int lds = STAGES * STAGE_BYTES;
  if (g_big_ablate && !out_f32) {
    switch (g_big_ablate) {
      case 1: return launch_big_ablate<1>(a, st);
      case 2: return launch_big_ablate<2>(a, st);
      case 3: return launch_big_ablate<3>(a, st);
      case 4: return launch_big_ablate<4>(a, st);
      case 7: return launch_big_ablate<7>(a, st);
      case 8: return launch_big_ablate<8>(a, st);
      case 11: return launch_big_ablate<11>(a, st);
      case 12: return launch_big_ablate<12>(a, st);
      case 16: return launch_big_ablate<16>(a, st);
      case 32: return launch_big_ablate<32>(a, st);
      case 48: return launch_big_ablate<48>(a, st);
      case 64: return launch_big_ablate<64>(a, st);
      case 192: return launch_big_ablate<192>(a, st);
      default: break;
    }
  }
  static bool attr_set = false;
  if (!attr_set) {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  if (out_f32)
    hipLaunchKernelGGL(conv_big_kernel<true>, dim3(a.total_tiles), dim3(512), lds, st, a);
  else
    hipLaunchKernelGGL(conv_big_kernel<false>, dim3(a.total_tiles), dim3(512), lds, st, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
