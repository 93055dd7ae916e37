// rn_conv_big_epi.h — per-wave epilogue shared by the 256 x 256 kernels (rn_conv_big.hip, rn_conv_halo.hip).
// Wave (wave_m, wave_n) of the 2 x 4 grid owns 128 pixels x 64 channels as 4 x 2 MFMA 32x32 accumulators:
//   acc[i][j][r]: pixel m = m0 + wave_m*128 + i*32 + (lane&31),
//                 channel n = n0 + wave_n*64 + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
// (weights = MFMA A, pixels = MFMA B, so a lane's registers are 4 consecutive channels of ONE pixel).
// bf16 output: raw accumulators rounded to bf16 (the reference's Conv2D output under the mixed policy is a bf16
// tensor), transposed through a 32 x 64 LDS patch (8-byte writes), read back 16 bytes = 8 channels per lane, then
// BatchNorm scale/shift (-> bf16), residual add (-> bf16), activation, full 128-byte row stores (the rounding
// points of rnet_hip.h's rn_conv_segment); optional fused BatchNorm forward statistics of the stored values.  f32 output: 32 x 32 f32 patches per j.  `patch` is this
// wave's 4 KB of LDS; no workgroup barrier.  The caller re-initialises acc for its next tile (big_acc_init).
#ifndef RN_CONV_BIG_EPI_H_
#define RN_CONV_BIG_EPI_H_
#include <type_traits>

#include "rn_conv_dev.h"

#ifdef HALO_PROF   // probe builds (tools/probes/build_halo_ablate.sh): cycle stamps of workgroup 0, thread 0
__device__ unsigned long long g_halo_clk[48];
// (unconditional store, no read-back: a load here would wait, through vmcnt, for every store issued before it)
#define EPI_STAMP(k_) if (blockIdx.x == 0 && threadIdx.x == 0) g_halo_clk[k_] = clock64();
#else
#define EPI_STAMP(k_)
#endif

#ifdef HALO_ABLATE
#define EPI_ABLATE HALO_ABLATE   // probe builds: 16 = (almost) no output stores
#else
#define EPI_ABLATE 0
#endif

typedef rn_h16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  bf16x2_t b = __builtin_convertvector(v, bf16x2_t);  // v_cvt_pk_bf16_f32: RNE, like rn_f32_to_bf16
  return __builtin_bit_cast(uint32_t, b);
}
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, a), __builtin_bit_cast(i16x2_t, b)));
}
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(i16x2_t, a), __builtin_bit_cast(i16x2_t, b)));
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return rn_lo16(u); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return rn_hi16(u); }

// XCD-aware tile numbering: virtual id v (dispatched round-robin over the 8 XCDs) -> tile, so that
// consecutive tiles (same pixels, neighbouring channel tiles) stay on one XCD's L2
__device__ __forceinline__ int tile_of(int v, int total) {
  const int xcd = v & 7, slot = v >> 3;
  const int q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// The Conv2D layer's bias is the accumulators' initial value, so the epilogue rounds acc + bias to bf16 ONCE (the
// layer's output tensor: fp32 BiasAdd inside the layer, then the cast) and bias-only layers (the head towers) need
// no fp32 pass over the transposed rows — relu / relu6 act on the packed bf16.
template <bool OUT_F32, bool HAS_RES>
__device__ __forceinline__ bool big_bias_in_acc(const ConvArgs& args, const ConvSegDev& sg) {
  // (compiled out of the residual variants: no layer of the reference has both a bias and a residual input — the
  // dispatcher keeps such launches on the 128-row kernel — and the extra code cost the <false, true> kernel 6 %)
  if (HAS_RES) return false;
  return sg.bias != nullptr;
}
// accumulators of a new tile: zero, or the bias of the lane's channels (see the layout at the top).  The 64
// floats of the wave come through the scalar cache (uniform address, constant address space): no vector memory
// operation, so the counted vmcnt waits of the DMA stream are not disturbed.
template <bool OUT_F32, bool HAS_RES, int WM = 2>   // WM: waves along the pixels (8 / WM along the channels)
__device__ __forceinline__ void big_acc_init(f32x16_t (&acc)[4][2], const ConvArgs& args, int c_si, int c_n0, int wave,
                                             bool with_bias = true) {   // false: parts 1.. of a split tile start from zero
  const ConvSegDev& sg = args.seg[c_si];
  if (OUT_F32 && with_bias && sg.pair_cout && sg.bias != nullptr) {
    // w_pair: accumulator tile j = 0 (hi plane) of the wave's column block starts from the bias of channels
    // [nw0 / 2, nw0 / 2 + 32), tile j = 1 (lo plane) from zero; Cout need only be a multiple of 4 here
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* b = (const cfloat*)(unsigned long long)sg.bias;
    const int cw0 = (c_n0 + (wave % (8 / WM)) * 64) >> 1;
    const int C = sg.pair_cout;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool hi = lane >= 32;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ca = cw0 + g * 8 + q, cb = ca + 4;
        const float lo_v = b[ca < C ? ca : 0], hi_v = b[cb < C ? cb : 0];   // channels past C are never stored
        const float v = hi ? hi_v : lo_v;
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i][0][g * 4 + q] = v; acc[i][1][g * 4 + q] = 0.0f; }
      }
    return;
  }
  if (with_bias && big_bias_in_acc<OUT_F32, HAS_RES>(args, sg)) {
    typedef const float __attribute__((address_space(4))) cfloat;
    const cfloat* b = (const cfloat*)(unsigned long long)sg.bias;
    const int nw0 = c_n0 + (wave % (8 / WM)) * 64;
    const int Cout = sg.Cout;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool hi = lane >= 32;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int c0 = nw0 + j * 32 + g * 8;               // 8 channels: lanes 0-31 hold c0..c0+3, lanes 32-63 c0+4..c0+7
        c0 = c0 + 8 <= Cout ? c0 : Cout - 8;         // groups past Cout are never stored: read valid memory
        c0 = c0 < 0 ? 0 : c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo_v = b[c0 + q], hi_v = b[c0 + 4 + q];
          const float v = hi ? hi_v : lo_v;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j][g * 4 + q] = v;
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }
}

// HAS_RES = false compiles the residual input out: the prefetched residual rows are global loads, and the waits
// the compiler puts in front of their uses also wait for the stores issued before them (loads and stores share
// vmcnt and retire in order) — with the loads merely predicated off, every 32-pixel block of a tile would still
// wait for the previous block's stores to reach L2 (measured: 25 000 cycles per tile epilogue instead of ~5 000).
// BN_BWD (bf16, no residual / affine / bias / activation on the launch): the launch writes dz of a BatchNorm + ReLU
// layer; the wave also reads that layer's raw conv output y for its rows (through the residual prefetch registers)
// and accumulates stage 1 of the backward reduction — sum g, sum g*xhat, g = dz_stored * [y*scale + shift > 0] — into
// bn_partial (rnet_hip.h: rn_conv_segment.bn_bwd_y).
// FROM_WS (part 0 of a split tile, rn_conv_halo.hip): the accumulators are not in registers but in the split-K workspace,
// one slot per part — [wave][32 x (64 lanes x 16 B)], piece (i*2 + j)*4 + q = registers 4q..4q+3 of acc[i][j] — and each
// 32-pixel block (i, j) is rebuilt, parts added in order, right before the epilogue consumes it: 16 - 32 live accumulator
// registers instead of 128 (rebuilding all of them up front spilled; a kernel with scratch pays per launch).
struct BigEpiSrc {
  const char* slots;   // first slot of the tile
  int nparts, voff;    // parts to add; this lane's byte offset inside a slot (wave * 32768 + lane * 16)
};
// NJ accumulator tiles (i, j0 .. j0 + NJ - 1) = the sum over the parts, in part order.  The slots were written with sc1 stores
// and are read back from memory (~2 us per round trip): every load of the block — 4 parts x NJ x 4 quads, the parts past
// nparts at an out-of-range offset, which the buffer load returns as zeros — is issued before the first add (loaded part by
// part behind its own wait, a tile's 16 dependent round trips were 35 us of the owner's time).
template <int NJ>
__device__ __forceinline__ void big_epi_rebuild(f32x16_t* t, const BigEpiSrc& src, int i, int j0) {
  typedef unsigned u32x4_t_ __attribute__((ext_vector_type(4)));
  // one descriptor over the tile's consecutive slots; part p = p * slot bytes further
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)src.slots, 0, src.nparts * RN_SPLITK_SLOT_BYTES, 0x00020000);
  u32x4_t_ v[4][NJ][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        v[p][j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, src.voff + p * RN_SPLITK_SLOT_BYTES,
                                                           ((i * 2 + j0 + j) * 4 + q) * 1024, 16);   // sc1
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a0 = __uint_as_float(v[0][j][q].x), a1 = __uint_as_float(v[0][j][q].y);
      float a2 = __uint_as_float(v[0][j][q].z), a3 = __uint_as_float(v[0][j][q].w);
#pragma unroll
      for (int p = 1; p < 4; ++p) {
        a0 += __uint_as_float(v[p][j][q].x); a1 += __uint_as_float(v[p][j][q].y);
        a2 += __uint_as_float(v[p][j][q].z); a3 += __uint_as_float(v[p][j][q].w);
      }
      t[j][q * 4] = a0; t[j][q * 4 + 1] = a1; t[j][q * 4 + 2] = a2; t[j][q * 4 + 3] = a3;
    }
}

template <bool OUT_F32, bool HAS_RES, bool BN_BWD = false, bool FROM_WS = false, int WM = 2>
__device__ __forceinline__ void big_epilogue(f32x16_t (&acc)[4][2], const ConvArgs& args, int c_si, int c_m0,
                                             int c_n0, int wave, char* patch, const BigEpiSrc src = BigEpiSrc(),
                                             int m_end = -1, int chunk0 = -1) {
  // m_end / chunk0 (conv_big_kernel's balanced tiles): the tile's rows end at m_end (< c_m0 + 256) and its 128-row blocks
  // of fused BatchNorm partial sums are numbered from chunk0; -1: whole tiles, blocks numbered by c_m0 / 128
  static_assert(!(BN_BWD && (OUT_F32 || HAS_RES)), "BN_BWD: plain bf16 launches only");
  constexpr bool LOADS = HAS_RES || BN_BWD;   // the epilogue prefetches a second [M][Cout] bf16 tensor
  const int wave_m = wave / (8 / WM), wave_n = wave % (8 / WM);   // WM x 8/WM waves of 128 pixels x 64 channels
  const ConvSegDev& sg = args.seg[c_si];
  const int Cout = sg.Cout, M = m_end >= 0 ? m_end : sg.M;
  const int nw0 = c_n0 + wave_n * 64;
  const int mw0 = c_m0 + wave_m * 128;
  // a fresh lane id, so that nothing the epilogue needs stays live across the main loop
  const int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int fr = elane & 31, fh = elane >> 5;
  const int rrow = elane >> 3, ru = elane & 7;   // read-back: 8 lanes per pixel row, 8 rows per pass
  if (!OUT_F32) {
    // bf16 patch: 32 pixels x 64 channels (128 B rows, 16-byte units swizzled by the pixel row)
    const int nr = nw0 + ru * 8;                 // this lane's 8 channels on the read-back side
    const bool nok = nr < Cout;
    float sc[8], sf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { sc[q] = 1.0f; sf[q] = 0.0f; }
    if (nok && sg.scale) {
      const float4 a = *(const float4*)(sg.scale + nr), b = *(const float4*)(sg.scale + nr + 4);
      sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    }
    if (nok && sg.shift) {
      const float4 a = *(const float4*)(sg.shift + nr), b = *(const float4*)(sg.shift + nr + 4);
      sf[0] = a.x; sf[1] = a.y; sf[2] = a.z; sf[3] = a.w; sf[4] = b.x; sf[5] = b.y; sf[6] = b.z; sf[7] = b.w;
    }
    float mu[BN_BWD ? 8 : 1], is[BN_BWD ? 8 : 1];
    if (BN_BWD) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { mu[q] = 0.0f; is[q] = 0.0f; }
    }
    if (BN_BWD && nok) {   // the launch itself has no affine: sc / sf carry the BatchNorm layer's scale / shift (mask only)
      const float* f0 = sg.bn_fwd + nr;
      const float* f1 = sg.bn_fwd + Cout + nr;
      {
        const float4 a = *(const float4*)f0, b = *(const float4*)(f0 + 4), c = *(const float4*)f1, d = *(const float4*)(f1 + 4);
        mu[0] = a.x; mu[1] = a.y; mu[2] = a.z; mu[3] = a.w; mu[4] = b.x; mu[5] = b.y; mu[6] = b.z; mu[7] = b.w;
        is[0] = c.x; is[1] = c.y; is[2] = c.z; is[3] = c.w; is[4] = d.x; is[5] = d.y; is[6] = d.z; is[7] = d.w;
      }
      const float* f2 = sg.bn_fwd + 2 * Cout + nr;
      const float* f3 = sg.bn_fwd + 3 * Cout + nr;
      const float4 a = *(const float4*)f2, b = *(const float4*)(f2 + 4), c = *(const float4*)f3, d = *(const float4*)(f3 + 4);
      sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
      sf[0] = c.x; sf[1] = c.y; sf[2] = c.z; sf[3] = c.w; sf[4] = d.x; sf[5] = d.y; sf[6] = d.z; sf[7] = d.w;
    }
    // Take the scale / shift registers through an empty asm: the compiler has to wait for these (conditional)
    // loads HERE.  Left to the first real use it re-waits inside every predicated store block further down, and a
    // vmcnt wait there also waits for the stores issued before it (loads and stores retire in order): each
    // 8-row pass then costs a full store round trip to L2 (measured ~3 000 cycles per 32-pixel block).
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(sc[q]), "+v"(sf[q]));
    if (BN_BWD) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(mu[q]), "+v"(is[q]));
    }
    EPI_STAMP(12);
    const bool has_res = HAS_RES && sg.residual != nullptr;
    const uint16_t* const side = BN_BWD ? sg.bn_y : sg.residual;   // the tensor the prefetch reads
    const bool has_side = BN_BWD ? true : has_res;
    // fused BatchNorm forward statistics (training, raw conv output): per-lane sums of the stored values
    const bool stats = sg.bn_partial != nullptr;
    float st0[8], st1[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) st0[q] = st1[q] = 0.0f;
    // this lane's output address at (block 0, pass 0); blocks / passes are uniform multiples of a row further on
    char* const ybase = (char*)sg.y + ((long long)(mw0 + rrow) * Cout + nr) * 2;
    const long long row_bytes = (long long)Cout * 2;
    const int act = args.act;
    const bool clamp_lo = act == RN_ACT_RELU || act == RN_ACT_RELU6, clamp_hi = act == RN_ACT_RELU6;
    const bool affine = !BN_BWD && (sg.scale != nullptr || sg.shift != nullptr);
    // conv (+bias) output only: the transposed bf16 rows are the result, up to relu / relu6 on the packed pairs
    const bool plain = BN_BWD || (!affine && !has_res && act != RN_ACT_SWISH);
    const bool round2 = affine && has_res;   // the BatchNorm output is a bf16 tensor before the residual add
    uint4 rv[2][LOADS ? 4 : 1];
#define BIG_RES_PREFETCH(buf_, i_)                                                                    \
_Pragma("unroll") for (int pass = 0; pass < (LOADS ? 4 : 0); ++pass) {                              \
  const int m = mw0 + (i_) * 32 + pass * 8 + rrow;                                                  \
  rv[buf_][pass] = make_uint4(0u, 0u, 0u, 0u);                                                      \
  if (has_side && nok && m < M) rv[buf_][pass] = *(const uint4*)(side + (long long)m * Cout + nr);  \
}
    // The block loop is compiled several times, for the modes a launch can be in — PLAIN (conv (+bias) output, relu / relu6
    // on the packed pairs at most) with or without the fused statistics, and the general arithmetic — and ONE uniform
    // branch per tile picks the copy.  With the mode tested inside the loop every 8-row pass ran ~9 scalar branches around
    // the code it did not need (swish, the affine path, the statistics): 16 passes x ~350 cycles made the epilogue of a
    // 256 x 256 tile 8 000 cycles, 8 % of a head-tower tile and half of a short-K 1x1 tile (round-4 probe; the stores
    // themselves need ~2 000).  The clamp of PLAIN is branch-free: bounds -32768 / 32767 are the identity on packed i16.
    const uint32_t lo2 = clamp_lo ? 0u : 0x80008000u, hi2 = clamp_hi ? RN_SIX_X2 : 0x7fff7fffu;
    // (mode flags: 0 = off, 1 = on, 2 = decided at run time — the one fully general copy)
    auto blocks = [&](auto plain_c, auto stats_c, auto round2_c, auto swish_c) __attribute__((always_inline)) {
      constexpr bool PLAIN = decltype(plain_c)::value;     // compile time: no affine / residual / swish arithmetic
      constexpr int STATS = decltype(stats_c)::value, ROUND2 = decltype(round2_c)::value, SWISH = decltype(swish_c)::value;
      BIG_RES_PREFETCH(0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        EPI_STAMP(13 + 2 * i);
        if (i + 1 < 4) BIG_RES_PREFETCH((i + 1) & 1, i + 1);
        if (FROM_WS) big_epi_rebuild<2>(&acc[i][0], src, i, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = j * 32 + g * 8 + fh * 4;   // channel inside the wave's 64
            uint2 pk;
            pk.x = pack2(acc[i][j][g * 4 + 0], acc[i][j][g * 4 + 1]);
            pk.y = pack2(acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
            // ds_write_b64 is serviced in contiguous 16-lane groups over a 128-byte bank row: rows fr and fr + 8 share
            // the 16-byte unit (same fr & 7), so the 8-byte half is flipped for rows 8-15 / 24-31 — 16 distinct 8-byte
            // chunks per group instead of a 2-way conflict (SQ_LDS_BANK_CONFLICT was 18-25 % of the LDS cycles of the
            // short-K 1x1 launches); the read-back swaps the halves back for odd passes
            *(uint2*)(patch + fr * 128 + (((nl >> 3) ^ (fr & 7)) << 4) + ((((nl >> 2) ^ (fr >> 3)) & 1) << 3)) = pk;
          }
        EPI_STAMP(14 + 2 * i);
        // read the block's four 8-row passes back before touching any of them: one LDS round trip per block instead
        // of four dependent ones (each pass used to wait for its own read, then for a scalar reload of the output
        // pointer, before its store could issue: ~600 cycles per pass of pure latency, 16 passes per tile)
        uint4 vb[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int row = pass * 8 + rrow;
          const uint4 t = *(const uint4*)(patch + row * 128 + ((ru ^ (row & 7)) << 4));
          vb[pass] = (pass & 1) ? make_uint4(t.z, t.w, t.x, t.y) : t;   // rows 8-15 / 24-31: halves were flipped
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int row = pass * 8 + rrow;
          const uint4 v = vb[pass];
          const int m = mw0 + i * 32 + row;
          if (m < M && nok) {
            uint4 ov;
            if (PLAIN) {
              // raw conv output (training: BatchNorm follows) or bias-initialised accumulators: the transposed
              // bf16 rows are the result, up to relu / relu6 on the packed pairs (see below)
              ov.x = pk_min_i16(pk_max_i16(v.x, lo2), hi2); ov.y = pk_min_i16(pk_max_i16(v.y, lo2), hi2);
              ov.z = pk_min_i16(pk_max_i16(v.z, lo2), hi2); ov.w = pk_min_i16(pk_max_i16(v.w, lo2), hi2);
            } else {
              const uint4 r4 = HAS_RES ? rv[i & 1][LOADS ? pass : 0] : make_uint4(0u, 0u, 0u, 0u);
              float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
              const float rr[8] = {bf_lo(r4.x), bf_hi(r4.x), bf_lo(r4.y), bf_hi(r4.y),
                                   bf_lo(r4.z), bf_hi(r4.z), bf_lo(r4.w), bf_hi(r4.w)};
              // f = the Conv2D layer's bf16 output; BatchNorm affine -> bf16 tensor -> residual add -> bf16 tensor
#pragma unroll
              for (int q = 0; q < 8; ++q) f[q] = f[q] * sc[q] + sf[q];
              if (ROUND2 == 1 || (ROUND2 == 2 && round2)) {
#pragma unroll
                for (int q = 0; q < 8; q += 2) {   // v_cvt_pk_bf16_f32 (RNE) and back
                  const uint32_t pq = pack2(f[q], f[q + 1]);
                  f[q] = bf_lo(pq); f[q + 1] = bf_hi(pq);
                }
              }
#pragma unroll
              for (int q = 0; q < 8; ++q) f[q] += rr[q];
              if (SWISH == 1 || (SWISH == 2 && act == RN_ACT_SWISH)) {
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = rn_swish(rn_rb(f[q]));
              }
              ov.x = pack2(f[0], f[1]); ov.y = pack2(f[2], f[3]); ov.z = pack2(f[4], f[5]); ov.w = pack2(f[6], f[7]);
              // relu / relu6 on the packed bf16 pairs: rounding is monotonic and 0 and 6 are bf16 values, so
              // act(round(x)) == round(act(x)); as signed 16-bit integers every negative bf16 is below 0 and
              // positive ones order like their values (one v_pk_max_i16 / v_pk_min_i16 per two outputs)
              ov.x = pk_min_i16(pk_max_i16(ov.x, lo2), hi2); ov.y = pk_min_i16(pk_max_i16(ov.y, lo2), hi2);
              ov.z = pk_min_i16(pk_max_i16(ov.z, lo2), hi2); ov.w = pk_min_i16(pk_max_i16(ov.w, lo2), hi2);
            }
            if (!(EPI_ABLATE & 16) || ov.x == 0x12345678u) *(uint4*)(ybase + (long long)(i * 32 + pass * 8) * row_bytes) = ov;
            if (STATS == 1 || (STATS == 2 && stats)) {
              const float w[8] = {bf_lo(ov.x), bf_hi(ov.x), bf_lo(ov.y), bf_hi(ov.y),
                                  bf_lo(ov.z), bf_hi(ov.z), bf_lo(ov.w), bf_hi(ov.w)};
              if (BN_BWD) {
                const uint4 y4 = rv[i & 1][LOADS ? pass : 0];
                const float yy[8] = {bf_lo(y4.x), bf_hi(y4.x), bf_lo(y4.y), bf_hi(y4.y),
                                     bf_lo(y4.z), bf_hi(y4.z), bf_lo(y4.w), bf_hi(y4.w)};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                  const float g = (yy[q] * sc[q] + sf[q]) > 0.0f ? w[q] : 0.0f;
                  st0[q] += g;
                  st1[q] += g * ((yy[q] - mu[BN_BWD ? q : 0]) * is[BN_BWD ? q : 0]);
                }
              } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) { st0[q] += w[q]; st1[q] += w[q] * w[q]; }
              }
            }
          }
        }
      }
    };
    typedef std::integral_constant<int, 0> off_t;
    typedef std::integral_constant<int, 1> on_t;
    typedef std::integral_constant<int, 2> runtime_t;
    if constexpr (BN_BWD) {   // always plain, always with the (backward) statistics: one copy
      blocks(std::true_type{}, on_t{}, off_t{}, off_t{});
    } else if (plain) {
      if (stats) blocks(std::true_type{}, on_t{}, off_t{}, off_t{});
      else blocks(std::true_type{}, off_t{}, off_t{}, off_t{});
    } else if (!stats && act != RN_ACT_SWISH) {
      // the inference layers (folded BatchNorm [+ residual] + relu) and the accumulating data gradients
      if (round2) blocks(std::false_type{}, off_t{}, on_t{}, off_t{});
      else blocks(std::false_type{}, off_t{}, off_t{}, off_t{});
    } else {
      blocks(std::false_type{}, runtime_t{}, runtime_t{}, runtime_t{});   // swish (EfficientNet), or statistics of a non-plain launch
    }
    EPI_STAMP(21);
    if (stats) {
      // sum over the 8 row lanes that share this lane's channels (lane bits 3..5), then lanes 0-7 write the
      // wave's 128-pixel row block: chunk = m_tile * 2 + wave_m, layout [chunk][2][Cout]
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
          st0[q] += __shfl_xor(st0[q], o, 64);
          st1[q] += __shfl_xor(st1[q], o, 64);
        }
      }
      if (rrow == 0 && nok) {
        float* dst = sg.bn_partial + ((long long)(chunk0 >= 0 ? chunk0 : c_m0 / 128) + wave_m) * 2 * Cout + nr;   // one row per 128 pixels
        *(float4*)(dst) = make_float4(st0[0], st0[1], st0[2], st0[3]);
        *(float4*)(dst + 4) = make_float4(st0[4], st0[5], st0[6], st0[7]);
        *(float4*)(dst + Cout) = make_float4(st1[0], st1[1], st1[2], st1[3]);
        *(float4*)(dst + Cout + 4) = make_float4(st1[4], st1[5], st1[6], st1[7]);
      }
    }
#undef BIG_RES_PREFETCH
  } else {
    // f32 output (prediction convs): f32 patch of 32 pixels x 32 channels per j (128 B rows)
    // w_pair: tile j = 1 holds the lo-plane products of tile j = 0's channels — one 32-channel block per wave,
    // channels [nw0 / 2, nw0 / 2 + 32) of a pair_cout-channel output
    const int pairC = sg.pair_cout;
    const int CoutY = pairC ? pairC : Cout;
    auto half = [&](auto j_c) __attribute__((always_inline)) {
      constexpr int j = decltype(j_c)::value;
      const int nr = (pairC ? (nw0 >> 1) : nw0 + j * 32) + ru * 4;      // this lane's 4 channels on the read-back side
      const bool nok = nr < CoutY;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nok && sg.scale) sc = *(const float4*)(sg.scale + nr);
      if (nok && sg.shift) sf = *(const float4*)(sg.shift + nr);
      asm volatile("" : "+v"(sc.x), "+v"(sc.y), "+v"(sc.z), "+v"(sc.w), "+v"(sf.x), "+v"(sf.y), "+v"(sf.z), "+v"(sf.w));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (FROM_WS) big_epi_rebuild<1>(&acc[i][j], src, i, j);
        if (j == 0 && pairC) {
          if (FROM_WS) big_epi_rebuild<1>(&acc[i][1], src, i, 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][0][r] += acc[i][1][r];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v;
          v.x = acc[i][j][g * 4 + 0]; v.y = acc[i][j][g * 4 + 1]; v.z = acc[i][j][g * 4 + 2]; v.w = acc[i][j][g * 4 + 3];
          *(float4*)(patch + fr * 128 + (((g * 2 + fh) ^ (fr & 7)) << 4)) = v;
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int row = pass * 8 + rrow;
          float4 v = *(const float4*)(patch + row * 128 + ((ru ^ (row & 7)) << 4));
          const int m = mw0 + i * 32 + row;
          if (m < M && nok) {
            const long long o = (long long)m * CoutY + nr;
            v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
            if (sg.residual) {
              const uint2 r2 = *(const uint2*)(sg.residual + o);
              v.x += bf_lo(r2.x); v.y += bf_hi(r2.x); v.z += bf_lo(r2.y); v.w += bf_hi(r2.y);
            }
            {
              float f4[4] = {v.x, v.y, v.z, v.w};
              rn_apply_act_n<4>(f4, args.act);
              v = make_float4(f4[0], f4[1], f4[2], f4[3]);
            }
            *(float4*)((float*)sg.y + o) = v;
          }
        }
      }
    };
    half(std::integral_constant<int, 0>{});
    if (!pairC) half(std::integral_constant<int, 1>{});
  }
}
#endif  // RN_CONV_BIG_EPI_H_
