// rn_conv_big.hip — persistent 256 x 256 x 32 implicit-GEMM conv kernel for layers with Cout >= 256.
//
// Why a second shape: the 128 x 128 tile of rn_conv.hip needs (128+128)*2 B of staged operands per
// 2*128*128 FLOP = 64 B/clk/CU at the MFMA peak, i.e. 100 % of the vector-memory -> LDS path, and every
// workgroup serialises load latency -> MFMA -> epilogue -> store (~2 TB/s on the
// HBM-bound 1x1 layers, 34 % of the MFMA peak on the 3x3 ones).  This kernel halves the staged bytes per
// FLOP and never lets the memory pipeline drain between tiles:
//
//   * one workgroup per CU, persistent: workgroup b walks tiles b, b+G, b+2G, ... (XCD-aware numbering)
//     and treats their K steps as ONE stream.  The LDS-DMA (`buffer_load ... lds`) runs 3-4 stream steps
//     ahead in a four-stage ring (4 x 32 KB), so the first operands of the next tile land while the
//     current tile runs its epilogue, and that tile's stores drain under the next tile's MFMAs.
//   * 512 threads = 8 wavefronts as 2 (M) x 4 (N), wave tile 128 pixels x 64 channels = 4 x 2 MFMA
//     32x32x16 tiles, 128 accumulator registers.  The two waves of a SIMD (wave w and w+4) ping-pong:
//     in every half step one runs its 16 MFMAs back to back from registers (compute segment) while the
//     other reads its 12 fragments of the next step, issues its 4 DMA pieces and waits for older pieces
//     (load segment); two s_barriers per K step.
//         slot 2g  : group 0 compute(g)               | group 1 load(read g, issue g+3)
//         slot 2g+1: group 0 load(read g+1, issue g+4) | group 1 compute(g)
//     Stream step T is first read in slot 2T-1 and its stage was last read (step T-4) in slot 2T-8; group 1
//     issues it in slot 2T-6, group 0 in slot 2T-7; at the END of a load segment a wave waits for everything
//     but the pieces of this and the previous segment (vmcnt 8) and the barrier that ends the segment
//     publishes them (group 1's pieces of T in slot 2T-2, group 0's in 2T-3): four slots of flight time.
//   * operands are swapped (weights = MFMA A, pixels = MFMA B), so a lane's accumulator registers are
//     4 consecutive output channels of ONE pixel.  Epilogue, per wave, no workgroup barrier: the raw
//     accumulators are rounded to bf16 (the reference's Conv2D output under the mixed policy is a bf16
//     tensor) and transposed through a 32 x 64 LDS patch (8-byte writes), read back 16 bytes = 8 channels
//     per lane, then scale/shift (folded BN + bias), residual add, activation in fp32, and full 128-byte
//     row stores.  The residual rows and the lane's 8 scale/shift values are prefetched before the
//     transposes.  Patches live in their own 32 KB of LDS (ring 128 KB + patches 32 KB = 160 KB).
#include "rn_conv_big_epi.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 32, NW = 8, STAGES = 4;
constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;
constexpr int RING_BYTES = STAGES * STAGE_BYTES, PATCH_BYTES = 4096;
constexpr int LDS_BYTES = RING_BYTES + NW * PATCH_BYTES;

template <bool OUT_F32, bool HAS_RES, bool BN_BWD = false>   // BN_BWD: rn_conv_big_epi.h
__global__ void __launch_bounds__(512) conv_big_kernel(const ConvArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total = args.total_tiles;
  const int G = gridDim.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 2, wave_n = wave & 3;   // wave_m is also the ping-pong group
  const int R = args.R, S = args.S, RS = R * S;

  // ---- issue side: the tile whose K steps are being DMA'd -----------------------------------------
  int i_v = blockIdx.x;           // virtual id of that tile; the stream ends when i_v >= total
  int i_tap = 0, i_c0 = 0;        // next K step of that tile
  int i_r = 0, i_s = 0;           // i_tap = i_r * S + i_s
  int g_iss = 0;                  // stream steps issued by this wave
  int i_W = 0, i_PS = 0, i_Cin = 0, i_cw = 0;   // i_cw: input channels wrap (split-bf16 weight planes)
  __amdgpu_buffer_rsrc_t rs_x, rs_w;
  unsigned a_off[2], a_mask[2], b_off[2];
  const int d_row = lane >> 2, d_pos = lane & 3;   // DMA instruction j fills rows (j*8 + wave)*16 + lane/4

#define BIG_SETUP_ISSUE()                                                                             \
  do {                                                                                                \
    const int tile__ = (args.pad_ & 2) ? i_v : tile_of(i_v, total); /* bit 1: dealt round-robin */     \
    int si__ = 0;                                                                                     \
    _Pragma("unroll 1") for (int i = 1; i < args.nseg; ++i)                                           \
      if (tile__ >= args.seg[i].tile_begin) si__ = i;                                                 \
    const ConvSegDev& sg__ = args.seg[si__];                                                          \
    const int lt__ = tile__ - sg__.tile_begin;                                                        \
    const int mt__ = rn_fdiv(lt__, sg__.n_tiles, rn_rcp((float)sg__.n_tiles)); /* < 2^22 tiles */  \
    const int rows_t__ = sg__.rows ? sg__.rows : BM;     /* pixels a tile covers (balanced tiles: < 256) */ \
    const int m0__ = mt__ * rows_t__, n0__ = (lt__ - mt__ * sg__.n_tiles) * BN;                       \
    const int mend__ = m0__ + rows_t__ < sg__.M ? m0__ + rows_t__ : sg__.M;                           \
    i_W = sg__.W; i_PS = sg__.pix_stride; i_Cin = sg__.CinP; i_cw = sg__.cwrap;                       \
    const int H__ = sg__.H, Ktot__ = RS * i_Cin;                                                      \
    const int rows__ = sg__.Cout <= 64 ? 64 : ((sg__.Cout + 127) / 128) * 128; /* packed weight rows */ \
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)sg__.x, 0,                                        \
                                             (int)((long long)sg__.N * H__ * i_W * i_PS * 2), 0x00020000); \
    rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)sg__.w, 0, (int)((long long)rows__ * Ktot__ * 2), \
                                             0x00020000);                                             \
    const float rWo__ = rn_rcp((float)sg__.Wo), rHo__ = rn_rcp((float)sg__.Ho);                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
      const int row = (j * NW + wave) * 16 + d_row;                                                   \
      const int chunk = d_pos ^ lds_swz<BK>(row);                                                     \
      const int m = m0__ + row;                                                                       \
      const int mm = m < mend__ ? m : 0;                                                              \
      int ox, oy, n;                                                                                  \
      if (args.pad_ & 1) {   /* every M < 2^22: float-reciprocal division (~8 VALU instead of ~45 each) */ \
        const int t2 = rn_fdiv(mm, sg__.Wo, rWo__);                                                   \
        ox = mm - t2 * sg__.Wo;                                                                       \
        n = rn_fdiv(t2, sg__.Ho, rHo__);                                                              \
        oy = t2 - n * sg__.Ho;                                                                        \
      } else {                                                                                        \
        ox = mm % sg__.Wo;                                                                            \
        const int t2 = mm / sg__.Wo;                                                                  \
        oy = t2 % sg__.Ho;                                                                            \
        n = t2 / sg__.Ho;                                                                             \
      }                                                                                               \
      const int iy0 = oy * args.sh - args.pt, ix0 = ox * args.sw - args.pl;                           \
      a_off[j] = (unsigned)(((((long long)n * H__ + iy0) * i_W + ix0) * i_PS + chunk * 8) * 2);       \
      unsigned mask = 0;                                                                              \
      if (m < mend__) {                                                                               \
        for (int r = 0; r < R; ++r)                                                                   \
          for (int s = 0; s < S; ++s) {                                                               \
            const bool ok = (unsigned)(iy0 + r) < (unsigned)H__ && (unsigned)(ix0 + s) < (unsigned)i_W; \
            mask |= (ok ? 1u : 0u) << (r * S + s);                                                    \
          }                                                                                           \
      }                                                                                               \
      a_mask[j] = mask;                                                                               \
      const int nrow = n0__ + row;                                                                    \
      b_off[j] = nrow < rows__ ? (unsigned)(((long long)nrow * Ktot__ + chunk * 8) * 2) : RN_OOB;     \
    }                                                                                                 \
  } while (0)

// DMA of the next stream step (4 pieces per wave); advances to the next tile when this one is done
#define BIG_ISSUE_STEP()                                                                              \
  do {                                                                                                \
    const int cw__ = i_c0 < i_cw ? i_c0 : (i_c0 < 2 * i_cw ? i_c0 - i_cw : i_c0 - 2 * i_cw);          \
    const unsigned tap_off__ = (unsigned)((((long long)i_r * i_W + i_s) * i_PS + cw__) * 2);          \
    const unsigned koff__ = (unsigned)(((long long)i_tap * i_Cin + i_c0) * 2);                        \
    char* st__ = smem + (g_iss & 3) * STAGE_BYTES;                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
      const unsigned v__ = ((a_mask[j] >> i_tap) & 1u) ? a_off[j] + tap_off__ : RN_OOB;               \
      dma16(rs_x, st__ + (j * NW + wave) * 1024, v__);                                                \
    }                                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
      dma16(rs_w, st__ + A_BYTES + (j * NW + wave) * 1024, b_off[j] == RN_OOB ? RN_OOB : b_off[j] + koff__); \
    ++g_iss;                                                                                          \
    i_c0 += BK;                                                                                       \
    if (i_c0 >= i_Cin) {                                                                              \
      i_c0 = 0;                                                                                       \
      ++i_tap;                                                                                        \
      if (++i_s == S) {                                                                               \
        i_s = 0;                                                                                      \
        ++i_r;                                                                                        \
      }                                                                                               \
      if (__builtin_expect(i_tap == RS, 0)) {                                                         \
        i_tap = i_r = i_s = 0;                                                                        \
        i_v += G;                                                                                     \
        if (i_v < total) BIG_SETUP_ISSUE();                                                           \
      }                                                                                               \
    }                                                                                                 \
  } while (0)

  // ---- fragment read offsets ---------------------------------------------------------------------
  // row = lane&31 of a 32-row tile, 16-byte slot = 2*kk + (lane>>5), XOR-swizzled by (row/4)&3; the
  // swizzle only depends on lane&31 because every tile starts at a multiple of 16 rows.
  int off_p0, off_w0;
  {
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 2) & 3;
    off_p0 = (wave_m * 128 + fr) * 64 + ((0 + fh) ^ sw) * 16;            // pixels, kk = 0
    off_w0 = A_BYTES + (wave_n * 64 + fr) * 64 + ((0 + fh) ^ sw) * 16;   // weights, kk = 0
  }

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  bf16x8_t px0[4], wt0[2], px1[4], wt1[2];   // fragments of one stream step: K slices 0..15 / 16..31
#define BIG_READ(stage_)                                                                              \
  do {                                                                                                \
    const char* b__ = smem + (stage_) * STAGE_BYTES;                                                  \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) wt0[j] = *(const bf16x8_t*)(b__ + off_w0 + j * 2048); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) px0[i] = *(const bf16x8_t*)(b__ + off_p0 + i * 2048); \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) wt1[j] = *(const bf16x8_t*)(b__ + (off_w0 ^ 32) + j * 2048); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) px1[i] = *(const bf16x8_t*)(b__ + (off_p0 ^ 32) + i * 2048); \
  } while (0)
// load segment: fragment reads first (their LDS latency runs under the DMA issue), then this wave's
// pieces of the stream step 3-4 ahead, then the counted wait
#define BIG_LOADSEG(stage_)                                                                           \
  do {                                                                                                \
    BIG_READ(stage_);                                                                                 \
    if (i_v < total) {                                                                                \
      BIG_ISSUE_STEP();                                                                               \
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                \
    } else {                                                                                          \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
    }                                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
  } while (0)
#define BIG_COMPUTESEG()                                                                              \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        acc[i][j] = RN_MFMA_32x32x16(wt0[j], px0[i], acc[i][j], 0, 0, 0);      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        acc[i][j] = RN_MFMA_32x32x16(wt1[j], px1[i], acc[i][j], 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#define BIG_BARRIER()                       \
  do {                                      \
    __builtin_amdgcn_s_barrier();           \
    asm volatile("" ::: "memory");          \
  } while (0)

  // ---- compute side: the tile being accumulated --------------------------------------------------
  int c_v = blockIdx.x;   // virtual id
  int c_k = 0;            // K steps of it already accumulated
  int c_ksteps, c_m0, c_n0, c_si, c_mend, c_chunk0;   // c_mend: end of the tile's rows; c_chunk0: its first 128-row block
#define BIG_SETUP_COMPUTE()                                                                           \
  do {                                                                                                \
    const int tile__ = (args.pad_ & 2) ? c_v : tile_of(c_v, total);                                   \
    c_si = 0;                                                                                         \
    _Pragma("unroll 1") for (int i = 1; i < args.nseg; ++i)                                           \
      if (tile__ >= args.seg[i].tile_begin) c_si = i;                                                 \
    const ConvSegDev& sg__ = args.seg[c_si];                                                          \
    const int lt__ = tile__ - sg__.tile_begin;                                                        \
    const int mt__ = rn_fdiv(lt__, sg__.n_tiles, rn_rcp((float)sg__.n_tiles)); /* < 2^22 tiles */  \
    const int rows_t__ = sg__.rows ? sg__.rows : BM;                                                  \
    c_m0 = mt__ * rows_t__;                                                                           \
    c_mend = c_m0 + rows_t__ < sg__.M ? c_m0 + rows_t__ : sg__.M;                                     \
    c_chunk0 = mt__ * 2;                                                                              \
    c_n0 = (lt__ - mt__ * sg__.n_tiles) * BN;                                                         \
    c_ksteps = RS * (sg__.CinP / BK);                                                                 \
    c_k = 0;                                                                                          \
  } while (0)

  // ---- epilogue of the compute-side tile (per wave; acc is re-initialised for the next tile afterwards) --------------------
  auto epilogue = [&]() __attribute__((always_inline)) {
    big_epilogue<OUT_F32, HAS_RES, BN_BWD>(acc, args, c_si, c_m0, c_n0, wave, smem + RING_BYTES + wave * PATCH_BYTES,
                                           BigEpiSrc(), c_mend, c_chunk0);
  };

  // ---- prologue ------------------------------------------------------------------------------------
  BIG_SETUP_ISSUE();
  BIG_SETUP_COMPUTE();
  big_acc_init<OUT_F32, HAS_RES>(acc, args, c_si, c_n0, wave);
#pragma unroll 1
  for (int t = 0; t < 3 && i_v < total; ++t) BIG_ISSUE_STEP();
  // stream steps 0 and 1 must be complete before the first slot
  if (g_iss >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BIG_BARRIER();
  // pre-roll: both groups read step 0 and issue step 3; group 1 does so as its slot-0 load segment
  BIG_LOADSEG(0);
  if (wave_m == 1) BIG_BARRIER();

  // ---- main loop over this workgroup's stream --------------------------------------------------------
  // One loop for both groups (two uniform branches around the barriers; two straight-line copies made the
  // kernel larger than the instruction cache and cost registers).  Both groups execute 2N barriers for N stream steps; a finished tile is written out in the odd slot by
  // both (group 1 right after its MFMAs, group 0 before its load segment).
#define BIG_TILE_END()                                       \
  if (__builtin_expect(c_k == c_ksteps, 0)) {               \
    epilogue();                                             \
    if (c_v + G >= total) break;                            \
    c_v += G;                                               \
    BIG_SETUP_COMPUTE();                                    \
    big_acc_init<OUT_F32, HAS_RES>(acc, args, c_si, c_n0, wave); \
  }
  int g = 0;
#pragma unroll 1
  while (true) {
    BIG_COMPUTESEG();
    ++c_k;
    if (wave_m == 0) BIG_BARRIER();
    BIG_TILE_END();
    if (wave_m == 1) BIG_BARRIER();
    ++g;
    BIG_LOADSEG(g & 3);   // past the end of the stream: stale reads, nothing issued
    BIG_BARRIER();
  }
#undef BIG_TILE_END
#undef BIG_SETUP_COMPUTE
#undef BIG_BARRIER
#undef BIG_COMPUTESEG
#undef BIG_LOADSEG
#undef BIG_READ
#undef BIG_ISSUE_STEP
#undef BIG_SETUP_ISSUE
}

}  // namespace

int rn_launch_conv_big(const ConvArgs& a, bool out_f32, const rn_launch_opts& opts, hipStream_t st) {
  static unsigned long long attr_set = 0;   // one bit per device
  if (RN_ATTRS_NEEDED(attr_set)) {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<false, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<false, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<true, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<true, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)conv_big_kernel<false, false, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RN_ATTRS_DONE(attr_set);
  }
  // one persistent workgroup per CU (minus the CUs kept for RCCL; opts.max_workgroups caps it)
  const int grid = rn_persistent_grid(a.total_tiles, rn_num_cus(), opts);
  bool has_res = false;   // one residual input anywhere -> the variant that carries the residual path
  for (int i = 0; i < a.nseg; ++i) has_res = has_res || a.seg[i].residual != nullptr;
  const dim3 g3(grid), b3(512);
  if (a.seg[0].bn_y) {   // data gradient + stage 1 of the BatchNorm backward reduction (validated by the caller)
    hipLaunchKernelGGL((conv_big_kernel<false, false, true>), g3, b3, LDS_BYTES, st, a);
  } else if (out_f32) {
    if (has_res) hipLaunchKernelGGL((conv_big_kernel<true, true>), g3, b3, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_big_kernel<true, false>), g3, b3, LDS_BYTES, st, a);
  } else {
    if (has_res) hipLaunchKernelGGL((conv_big_kernel<false, true>), g3, b3, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv_big_kernel<false, false>), g3, b3, LDS_BYTES, st, a);
  }
  RN_CHECK_LAUNCH();
  return RN_OK;
}
