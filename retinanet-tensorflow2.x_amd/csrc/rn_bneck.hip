// rn_bneck.hip — one launch per bottleneck block of ResNet stage 1 in inference form (round 6).
//
// retinanet/model/backbone/resnet.py:194-248 `bottleneck_block` with filters = 64:
//     a   = relu(BN(conv1x1  Cx -> 64 (x)))
//     b   = relu(BN(conv3x3  64 -> 64 (a)))                      stride 1 (block_group1, :324-331), zero padding 1
//     out = relu(BN(conv1x1  64 -> 256 (b)) + shortcut)          shortcut = x (Cx = 256) or BN(conv1x1 64 -> 256 (x)) (Cx = 64)
// with every BatchNorm folded to scale / shift (moving statistics): the `resnet_initial` layers frozen by the 3x
// configs run like this in training (executor.py:154-176), every layer does when serving.
//
// Why one kernel.  As three (four) launches the block moves x -> a -> b -> out through HBM: at 160 x 160 x 32 images that is
// 420 + 105 | 105 + 105 | 105 + 420 + 420 = 1 680 MB for 114 GFLOP — each launch HBM-bound (AI 28 - 288 FLOP/B), 432 us
// per block.  Fused, a and b never leave the chip: 840 MB, the block's own input and output.
//
// How.  A workgroup owns full-width rows of ONE image: W = 32 * NW pixels, wave w owns the 32-pixel column block w, and
// the workgroup walks its row segment top to bottom.  Per row and wave three chained MFMA GEMMs (v_mfma_f32_32x32x16,
// weights = A operand, the wave's 32 pixels = B operand, so a lane's accumulators are channels of ONE pixel):
//   A  a(row r + 2)   = Wa . x          x fragments straight from global memory: lane (pixel n, half h) loads the 16 bytes
//                                       [16 s + 8 h, + 8) of its pixel for K step s — no LDS staging; producer waves run
//                                       this stage one row ahead of the consumers (below), four K chunks in flight;
//                                       result -> bf16 -> BN -> bf16 -> relu -> the `a` ring in LDS (SLOTS rows of W + 2
//                                       pixels x 128 B, 16-byte units XOR-swizzled by the pixel index: conflict-free
//                                       ds_write_b128 / ds_read_b128), one zero pixel on either side, zero rows outside the image
//   B  b(row r)       = sum over 9 taps Wb[tap] . a(shifted)    Wb resident in LDS (72 KB, fragment-major: a linear
//                                       ds_read_b128 per fragment); pixel fragments are 16-byte reads of the ring
//   C  out(row r)     = Wo . b (+ Wsc . x)   b NEVER goes through LDS: the MFMA output rows are permuted (bits 2 and 3 of
//                                       the row index swapped in the packed weights) so that a lane's accumulator
//                                       registers 8k .. 8k + 7 are the 8 CONSECUTIVE channels 16k + 8h + i of its pixel —
//                                       exactly the B-operand fragment of K step k of the next GEMM.  The output (and the
//                                       identity shortcut's input) passes through a 4 KB per-wave LDS patch, 64 channels
//                                       at a time, so that global memory sees full 128-byte rows (consumer waves below)
// Rounding points are the unfused path's (rnet_hip.h rn_conv_segment): Conv2D output -> bf16, BatchNorm -> bf16, (+ shortcut)
// -> relu -> bf16.  K orders differ from the unfused kernels, so results agree to the last fp32 bit of the accumulations,
// not bit for bit (tests: <= 1 bf16 ulp against the three-launch path on all but a few 1e-4 of the elements, and the
// float64 reference).
//
// Producer waves (one per column block) run stage A one row ahead; consumer waves (one per column block) run stages B and
// C; they meet at one or two LDS-only workgroup barriers per row.
//
// The work is HBM-bound by design: 139 KFLOP per pixel = 136 MFMAs per wave and row (~4 400 cycles) against ~17 000 cycles
// of HBM time per row at one workgroup per CU; LDS = 72 KB (Wb) + 5 KB (folded BatchNorm vectors) + ring + 4 KB per consumer
// wave <= 160 KB allows W <= 160 (3 ring rows and two barriers per row there; 4 rows and one barrier up to W = 128).
// Measured (tools/probes/bneck_ablate.sh): it is bound by L2 <-> CU transactions — Wa / Wo / Wsc fragments
// are re-read per wave and row because only Wb fits the LDS next to the ring — at 2.4 TB/s of algorithmic bytes: 1.26x (identity)
// / 1.85x (projection) the three-launch path at 32 images, 1.8x / 2.3x at 8, ~2x / ~3x at one.
#include "rn_conv_dev.h"

namespace {

typedef rn_h16 h16x2_t __attribute__((ext_vector_type(2)));
typedef float fx2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t bn_pack2(float lo, float hi) {
  fx2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2_t));   // v_cvt_pk_*: round to nearest even
}
__device__ __forceinline__ uint32_t bn_relu2(uint32_t a) {   // relu on a packed pair: negative 16-bit floats are negative int16
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, a), s16x2_t{0, 0}));
}

struct BneckArgs {
  const uint16_t* x;
  uint16_t* y;
  const uint16_t* wa;    // fragments [Cx / 16][2][64 lanes][8]
  const uint16_t* wb;    // fragments [9 taps * 4][2][64][8]                      73 728 bytes
  const uint16_t* wo;    // fragments [4][8][64][8]
  const uint16_t* wsc;   // fragments [4][8][64][8] (projection shortcut) or null
  const float* affine;   // a_scale[64] a_shift[64] b_scale[64] b_shift[64] o_scale[256] o_shift[256] (sc_scale[256] sc_shift[256])
  int N, H, W, Cx, rows_per_wg, segs;
};

constexpr int WB_BYTES = 9 * 4 * 2 * 64 * 16;   // 73 728
constexpr int AFF_FLOATS = 4 * 64 + 4 * 256;    // the affine vectors, copied to LDS (5 KB; the sc part only for Cx = 64)
constexpr int AFF_BYTES = AFF_FLOATS * 4;
// Workgroup barrier for LDS hand-offs only: the waves exchange nothing through global memory, so the prefetched x / weight
// loads stay in flight across it (__syncthreads would drain vmcnt too)
#define BN_BARRIER()                                        \
  do {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
  } while (0)

// weight fragment `frag` (1 KB each) of a packed matrix: buffer load with the fragment offset in an SGPR and lane * 16 as the
// only vector offset (flat loads made the compiler keep a 64-bit address pair per 4 KB of weights, hoisted and spilled)
typedef unsigned bn_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t bneck_wfrag(__amdgpu_buffer_rsrc_t rs, int frag, int lane16) {
  return __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, frag * 1024, 0));
}

// MFMA output row m of a 32-channel tile carries channel perm(m): bits 2 and 3 swapped (see the header)
__host__ __device__ inline int bneck_perm(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// 8 accumulator registers (channels c0 .. c0 + 7 of one pixel) -> Conv2D output (bf16) -> BatchNorm (bf16) -> relu, packed
__device__ __forceinline__ uint4 bneck_bn_relu(const f32x16_t& acc, int k, const float* __restrict__ scale,
                                               const float* __restrict__ shift, int c0) {
  // (scale / shift live in LDS: two lanes' worth of distinct addresses per read, broadcast)
  const float4 s0 = *(const float4*)(scale + c0), s1 = *(const float4*)(scale + c0 + 4);
  const float4 t0 = *(const float4*)(shift + c0), t1 = *(const float4*)(shift + c0 + 4);
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float sf[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = rn_rb(acc[8 * k + i]) * sc[i] + sf[i];
  uint4 o;
  o.x = bn_relu2(bn_pack2(f[0], f[1])); o.y = bn_relu2(bn_pack2(f[2], f[3]));
  o.z = bn_relu2(bn_pack2(f[4], f[5])); o.w = bn_relu2(bn_pack2(f[6], f[7]));
  return o;
}

// DBG: timing probes, compiled into probe builds only (-DRN_PROBES, rn_launch_opts.ablate picks one; tools/bench_bneck.py):
// 1 no residual loads, 2 no stores, 4 no Wo loads, 8 no x loads in stage A, 32 no stage B, 64 no Wa loads —
// wrong results by construction.  The product build has DBG = 0 only.
template <int CX, int SLOTS, int DBG = 0>
__global__ void __launch_bounds__(768) bneck64_kernel(const BneckArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KSA = CX / 16;          // K steps of stage A
  constexpr bool PROJ = CX == 64;       // projection shortcut (the block's input has 64 channels)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = args.W >> 5;           // column blocks = consumer waves = producer waves
  const bool producer = wave_id >= NW;
  const int wave = producer ? wave_id - NW : wave_id;   // column block
  const int n = lane & 31, h = lane >> 5;
  const int W = args.W, H = args.H;
  const int img = blockIdx.x / args.segs, seg = blockIdx.x - img * args.segs;
  const int r0 = seg * args.rows_per_wg;
  const int r1 = r0 + args.rows_per_wg < H ? r0 + args.rows_per_wg : H;
  if (r0 >= H) return;
  const int rowb = (W + 2) * 128;       // bytes of one ring row
  char* const wb_lds = smem;
  const float* const aff = (const float*)(smem + WB_BYTES);
  char* const ring = smem + WB_BYTES + AFF_BYTES;
  const int px = wave * 32 + n;         // this lane's pixel column
  const int pp = px + 1;                // ... in ring coordinates (zero column on either side)
  // The folded BatchNorm vectors are loop-invariant LDS reads: the compiler hoisted hundreds of them out of the row loop
  // and spilled.  `opq` is an opaque zero refreshed per stage, so every read stays where it is used.
  int opq = 0;
#define BN_OPAQUE() asm volatile("v_mov_b32 %0, 0" : "=v"(opq))
#define a_scale (aff + opq)
#define a_shift (aff + 64 + opq)
#define b_scale (aff + 128 + opq)
#define b_shift (aff + 192 + opq)
#define o_scale (aff + 256 + opq)
#define o_shift (aff + 512 + opq)
#define s_scale (aff + 768 + opq)
#define s_shift (aff + 1024 + opq)

  // ---- prologue: Wb and the affine vectors -> LDS, zero border columns ------------------------------------------
  for (int i = tid; i < WB_BYTES / 16; i += blockDim.x) ((uint4*)wb_lds)[i] = ((const uint4*)args.wb)[i];
  for (int i = tid; i < (PROJ ? AFF_FLOATS : AFF_FLOATS - 512) / 4; i += blockDim.x)
    ((float4*)(smem + WB_BYTES))[i] = ((const float4*)args.affine)[i];
  for (int i = tid; i < SLOTS * 2 * 8; i += blockDim.x) {
    const int slot = i / 16, side = (i >> 3) & 1, u = i & 7;
    *(uint4*)(ring + slot * rowb + (side ? (W + 1) * 128 : 0) + u * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  const long long img_px = (long long)img * H * W;
  const int lane16 = lane * 16;
  const bf16x8_t bfrag_zero = __builtin_bit_cast(bf16x8_t, make_uint4(0u, 0u, 0u, 0u));
  auto slot_of = [&](int row) { return ((row % SLOTS) + SLOTS) % SLOTS; };
  BN_BARRIER();                         // Wb, the affine vectors and the zero columns are in place

  if (producer) {
    // ================= producer waves: stage A, one row ahead of what the consumers read ==============================
    const __amdgpu_buffer_rsrc_t rs_wa = __builtin_amdgcn_make_buffer_rsrc((void*)args.wa, 0, KSA * 2 * 1024, 0x00020000);
    // (Until the round's last profile a row was "warmed" into L2 two rows ahead with one 4-byte load per line.  The FETCH_SIZE
    // pass showed what that did — x crossed the fabric THREE times per launch, 1.3 GB for a 420 MB tensor: a CU streams 160 KB
    // per row, 32 CUs share a 4 MB L2, so a warmed line is gone again before stage A reads it — and the ablation probe had
    // already said the warm loads bought nothing (342.7 vs 346.0 us): the producers have slack, their chunk pipeline simply
    // reads x from HBM.  Removed.)
    auto stage_a = [&](int row) __attribute__((always_inline)) {
      char* const dst = ring + slot_of(row) * rowb + pp * 128;
      const int sw = (pp >> 1) & 7;
      if (row < 0 || row >= H) {        // rows outside the image: the 3x3 conv's zero padding
#pragma unroll
        for (int u = h; u < 8; u += 2) *(uint4*)(dst + ((u ^ sw) << 4)) = make_uint4(0u, 0u, 0u, 0u);
        return;
      }
      const uint16_t* const xp = args.x + (img_px + (long long)row * W + px) * CX + h * 8;
      BN_OPAQUE();
      f32x16_t acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
      // K steps in chunks of two (2 pixel + 4 weight fragments, L2 hits), loaded DEPTH chunks ahead of their MFMAs
      constexpr int NCH = KSA / 2, DEPTH = NCH < 4 ? NCH : 4;
      bf16x8_t xq[DEPTH][2], wq[DEPTH][4];
      auto load_chunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) xq[c % DEPTH][q] = (DBG & 8) ? bfrag_zero : *(const bf16x8_t*)(xp + (c * 2 + q) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) wq[c % DEPTH][q] = (DBG & 64) ? bfrag_zero : bneck_wfrag(rs_wa, c * 4 + q, lane16);
      };
#pragma unroll
      for (int c = 0; c < DEPTH - 1; ++c) load_chunk(c);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c + DEPTH - 1 < NCH) load_chunk(c + DEPTH - 1);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j] = RN_MFMA_32x32x16(wq[c % DEPTH][q * 2 + j], xq[c % DEPTH][q], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int c0 = j * 32 + 16 * k + 8 * h;
          const uint4 o = bneck_bn_relu(acc[j], k, a_scale, a_shift, c0);
          *(uint4*)(dst + (((c0 >> 3) ^ sw) << 4)) = o;
        }
    };
#pragma unroll 1
    for (int row = r0 - 1; row <= r0 + 1; ++row) stage_a(row);
    BN_BARRIER();                       // rows r0 - 1 .. r0 + 1 of a are in the ring
#pragma unroll 1
    for (int r = r0; r < r1; ++r) {
      if (SLOTS == 3) BN_BARRIER();     // the consumers have read row r - 1: its slot is the one a(r + 2) goes to
      if (r + 1 < r1) stage_a(r + 2);
      BN_BARRIER();                     // a(r + 2) is complete before stage B of row r + 1
    }
  } else {
    // ================= consumer waves: stage B (3x3 from the ring) and stage C (1x1 + shortcut + store) ===============
    const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc((void*)args.wo, 0, 32 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ws =
        __builtin_amdgcn_make_buffer_rsrc((void*)(PROJ ? args.wsc : args.wo), 0, 32 * 1024, 0x00020000);
    // Global memory sees the output (and the identity shortcut's input) only as FULL 128-byte rows.  In MFMA layout a lane
    // owns 16-byte pieces of one pixel, so a store instruction would write 32 pieces of 32 bytes in 32 different lines, every
    // line by four different instructions: the ablation probes (tools/probes/bneck_ablate.sh) priced that at 122 of 382 us
    // for the stores and 63 us for the residual loads.  So 64 channels (two output tiles) at a time go through this wave's
    // 4 KB LDS patch [32 pixels][8 x 16 B, unit ^ ((pixel >> 1) & 7)]: the residual comes in as four coalesced row loads (8
    // lanes per pixel line) and is read back as fragments, the results are written as fragments and leave as four coalesced
    // row stores.  LDS operations of one wave execute in order: no barrier.
    char* const patch = ring + SLOTS * rowb + wave * 4096;
    const int cpx = lane >> 3, cun = lane & 7;       // coalesced side: pixel 8q + cpx of the wave's block, 16-byte unit cun
    BN_BARRIER();                       // (the producers' first three rows)
#pragma unroll 1
    for (int r = r0; r < r1; ++r) {
      const long long rowpix = img_px + (long long)r * W + wave * 32;
      const uint16_t* const xr = args.x + (rowpix + n) * CX + 8 * h;      // fragment view of x(r) (projection shortcut)
      const char* const xrow = (const char*)(args.x + (rowpix + cpx) * CX) + cun * 16;   // coalesced view, + q * 8 pixels
      char* const yrow = (char*)(args.y + (rowpix + cpx) * 256) + cun * 16;
      // stage C's operands that do not depend on stage B — the first weight fragments, the shortcut's inputs — are requested
      // BEFORE stage B: their latency runs under its 72 MFMAs
      bf16x8_t wo_f[2][4], ws_f[PROJ ? 2 : 1][PROJ ? 4 : 1];
      uint4 rin[PROJ ? 1 : 2][PROJ ? 1 : 4];         // the residual rows of a 64-channel group, coalesced layout
      bf16x8_t xs[PROJ ? 4 : 1];
      auto load_tile = [&](int jo) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          wo_f[jo & 1][s] = (DBG & 4) ? bfrag_zero : bneck_wfrag(rs_wo, s * 8 + jo, lane16);
        if (PROJ) {
#pragma unroll
          for (int s = 0; s < 4; ++s) ws_f[PROJ ? jo & 1 : 0][PROJ ? s : 0] = bneck_wfrag(rs_ws, s * 8 + jo, lane16);
        }
      };
      auto load_res = [&](int m) __attribute__((always_inline)) {   // channels 64 m .. 64 m + 63 of the wave's 32 pixels
        if (!PROJ) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            rin[PROJ ? 0 : m & 1][PROJ ? 0 : q] =
                (DBG & 1) ? make_uint4(0u, 0u, 0u, 0u) : *(const uint4*)(xrow + (long long)q * 8 * CX * 2 + m * 128);
        }
      };
      if (PROJ) {
#pragma unroll
        for (int s = 0; s < 4; ++s) xs[PROJ ? s : 0] = *(const bf16x8_t*)(xr + s * 16);
      }
      load_res(0);
      load_tile(0);
      // ---- stage B: b(r) = conv3x3(a) ----------------------------------------------------------------------------
      f32x16_t accb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) accb[j][q] = 0.0f;
      if (!(DBG & 32))
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const char* const rowp = ring + slot_of(r + dy - 1) * rowb;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int p = px + dx;        // ring pixel of image column px + dx - 1
          const char* const pix = rowp + p * 128;
          const int sw = (p >> 1) & 7;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bf16x8_t af = *(const bf16x8_t*)(pix + (((2 * t + h) ^ sw) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const bf16x8_t wf = *(const bf16x8_t*)(wb_lds + ((((dy * 3 + dx) * 4 + t) * 2 + j) * 64 + lane) * 16);
              accb[j] = RN_MFMA_32x32x16(wf, af, accb[j], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // one tap's 12 fragment reads at a time
        }
      }
      if (SLOTS == 3) BN_BARRIER();     // every consumer has read row r - 1
      // ---- stage C: out(r) = relu(BN(Wo . b) + shortcut) -----------------------------------------------------------
      bf16x8_t bfrag[4];
      BN_OPAQUE();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const uint4 o = bneck_bn_relu(accb[j], k, b_scale, b_shift, j * 32 + 16 * k + 8 * h);
          bfrag[2 * j + k] = __builtin_bit_cast(bf16x8_t, o);
        }
      const int psw = (n >> 1) & 7;                               // fragment side of the patch: pixel n
      char* const pfrag = patch + n * 128;
#pragma unroll
      for (int m = 0; m < 4; ++m) {     // 64 output channels per pass: tiles 2m, 2m + 1
        uint4 rfr[2][2];
        if (!PROJ) {
          // the residual rows of this group: patch <- coalesced registers, fragments <- patch; then the next group's loads
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(uint4*)(patch + (q * 8 + cpx) * 128 + ((cun ^ (((q * 8 + cpx) >> 1) & 7)) << 4)) = rin[PROJ ? 0 : m & 1][PROJ ? 0 : q];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k) rfr[t][k] = *(const uint4*)(pfrag + (((t * 4 + 2 * k + h) ^ psw) << 4));
          if (m + 1 < 4) load_res(m + 1);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int jo = 2 * m + t;
          BN_OPAQUE();
          if (jo + 1 < 8) load_tile(jo + 1);
          f32x16_t acco, accs;
#pragma unroll
          for (int q = 0; q < 16; ++q) { acco[q] = 0.0f; accs[q] = 0.0f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) acco = RN_MFMA_32x32x16(wo_f[jo & 1][s], bfrag[s], acco, 0, 0, 0);
          if (PROJ) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              accs = RN_MFMA_32x32x16(ws_f[PROJ ? jo & 1 : 0][PROJ ? s : 0], xs[PROJ ? s : 0], accs, 0, 0, 0);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int c0 = jo * 32 + 16 * k + 8 * h;
            const float4 s0 = *(const float4*)(o_scale + c0), s1 = *(const float4*)(o_scale + c0 + 4);
            const float4 t0 = *(const float4*)(o_shift + c0), t1 = *(const float4*)(o_shift + c0 + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float sf[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            float f[8], rr[8];
            if (PROJ) {   // the shortcut is a bf16 tensor of its own: Conv2D output -> bf16 -> BatchNorm -> bf16
              const float4 u0 = *(const float4*)(s_scale + c0), u1 = *(const float4*)(s_scale + c0 + 4);
              const float4 v0 = *(const float4*)(s_shift + c0), v1 = *(const float4*)(s_shift + c0 + 4);
              const float qs[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
              const float qf[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
              for (int i = 0; i < 8; ++i) rr[i] = rn_rb(rn_rb(accs[8 * k + i]) * qs[i] + qf[i]);
            } else {
              const uint4 r4 = rfr[t][k];
              rr[0] = rn_lo16(r4.x); rr[1] = rn_hi16(r4.x); rr[2] = rn_lo16(r4.y); rr[3] = rn_hi16(r4.y);
              rr[4] = rn_lo16(r4.z); rr[5] = rn_hi16(r4.z); rr[6] = rn_lo16(r4.w); rr[7] = rn_hi16(r4.w);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = rn_rb(rn_rb(acco[8 * k + i]) * sc[i] + sf[i]) + rr[i];
            uint4 o;
            o.x = bn_relu2(bn_pack2(f[0], f[1])); o.y = bn_relu2(bn_pack2(f[2], f[3]));
            o.z = bn_relu2(bn_pack2(f[4], f[5])); o.w = bn_relu2(bn_pack2(f[6], f[7]));
            *(uint4*)(pfrag + (((t * 4 + 2 * k + h) ^ psw) << 4)) = o;      // (the residual fragments were all read above)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // the group's 32 x 128 bytes leave as rows
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 o = *(const uint4*)(patch + (q * 8 + cpx) * 128 + ((cun ^ (((q * 8 + cpx) >> 1) & 7)) << 4));
          if (!(DBG & 2) || o.x == 0x12345678u) *(uint4*)(yrow + (long long)q * 8 * 256 * 2 + m * 128) = o;
        }
      }
      BN_BARRIER();                     // end of the row: a(r + 2) is complete, row r - 1's slot may be reused
    }
  }
#undef a_scale
#undef a_shift
#undef b_scale
#undef b_shift
#undef o_scale
#undef o_shift
#undef s_scale
#undef s_shift
#undef BN_OPAQUE
}

// ---- weight packing: f32 HWIO master kernels -> bf16 (f16) MFMA fragments ------------------------------------------------
// fragment element (kstep, tile, lane, i) = W[cin = 16 * kk + 8 * (lane >> 5) + i][cout = 32 * tile + perm(lane & 31)],
// kk = kstep for the 1x1 layers; for the 3x3 layer kstep = tap * 4 + kk (tap = kh * 3 + kw)
__global__ void bneck_pack_kernel(const float* __restrict__ w, int taps, int cin, int cout, uint16_t* __restrict__ dst) {
  const int ksteps = taps * (cin / 16), tiles = cout / 32;
  const long long total = (long long)ksteps * tiles * 64 * 8;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const long long q = e >> 9;
    const int tile = (int)(q % tiles), kstep = (int)(q / tiles);
    const int tap = kstep / (cin / 16), kk = kstep % (cin / 16);
    const int ci = 16 * kk + 8 * (lane >> 5) + i, co = 32 * tile + bneck_perm(lane & 31);
    dst[e] = rn_f32_to_bf16(w[((long long)tap * cin + ci) * cout + co]);
  }
}

size_t bneck_lds_bytes(int W, int slots) {   // Wb | affine vectors | ring | one 4 KB transpose patch per consumer wave
  return (size_t)WB_BYTES + AFF_BYTES + (size_t)slots * (W + 2) * 128 + (size_t)(W / 32) * 4096;
}
int bneck_slots(int W) {   // ring rows: 4 (one barrier per row) while the LDS holds them, else 3
  if (bneck_lds_bytes(W, 4) <= 160 * 1024) return 4;
  if (bneck_lds_bytes(W, 3) <= 160 * 1024) return 3;
  return 0;
}

size_t bneck_frag_elems(int taps, int cin, int cout) { return (size_t)taps * (cin / 16) * (cout / 32) * 512; }

}  // namespace

extern "C" int rn_bottleneck64_supported(int N, int H, int W, int Cx) {
  if (N <= 0 || H <= 0 || W <= 0 || (Cx != 64 && Cx != 256)) return 0;
  if (W % 32 != 0 || W / 32 > 6) return 0;   // 2 x (W / 32) waves per workgroup, at most 12
  return bneck_slots(W) ? 1 : 0;
}

// packed layout: [wa | wb | wo | wsc (Cx == 64 only)], every part a multiple of 1 KB
extern "C" size_t rn_bottleneck64_packed_bytes(int Cx) {
  if (Cx != 64 && Cx != 256) return 0;
  return 2 * (bneck_frag_elems(1, Cx, 64) + bneck_frag_elems(9, 64, 64) + bneck_frag_elems(1, 64, 256) +
              (Cx == 64 ? bneck_frag_elems(1, 64, 256) : 0));
}

extern "C" int rn_bottleneck64_pack(const float* wa_hwio, const float* wb_hwio, const float* wo_hwio,
                                    const float* wsc_hwio, int Cx, void* packed, void* stream) {
  RN_CHECK_ARG(Cx == 64 || Cx == 256, "rn_bottleneck64_pack: Cx = %d (64 | 256)", Cx);
  RN_CHECK_ARG(wa_hwio && wb_hwio && wo_hwio && packed, "rn_bottleneck64_pack: null pointer");
  RN_CHECK_ARG((Cx == 64) == (wsc_hwio != nullptr), "rn_bottleneck64_pack: projection kernel iff Cx == 64");
  uint16_t* dst = (uint16_t*)packed;
  hipStream_t st = (hipStream_t)stream;
  const struct { const float* w; int taps, cin, cout; } parts[4] = {
      {wa_hwio, 1, Cx, 64}, {wb_hwio, 9, 64, 64}, {wo_hwio, 1, 64, 256}, {wsc_hwio, 1, 64, 256}};
  for (int q = 0; q < 4; ++q) {
    if (!parts[q].w) continue;
    const size_t n = bneck_frag_elems(parts[q].taps, parts[q].cin, parts[q].cout);
    hipLaunchKernelGGL(bneck_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, parts[q].w, parts[q].taps,
                       parts[q].cin, parts[q].cout, dst);
    RN_CHECK_LAUNCH();
    dst += n;
  }
  return RN_OK;
}

extern "C" int rn_bottleneck64_fwd(const rn_bottleneck64_problem* p, void* stream) {
  RN_CHECK_ARG(p && p->x && p->y && p->w_packed && p->affine, "rn_bottleneck64_fwd: null pointer");
  RN_CHECK_ARG(rn_bottleneck64_supported(p->N, p->H, p->W, p->Cx) == 1,
               "rn_bottleneck64_fwd: unsupported shape N=%d H=%d W=%d Cx=%d (W %% 32 == 0, W <= 160, Cx 64 | 256)",
               p->N, p->H, p->W, p->Cx);
  RN_CHECK_ARG(rn_validate_launch_opts(p->opts, "rn_bottleneck64_fwd") == 0, "rn_bottleneck64_fwd: bad launch options");
  const int slots = bneck_slots(p->W);
  const size_t lds = bneck_lds_bytes(p->W, slots);
  static unsigned long long attr_set = 0;
  if (RN_ATTRS_NEEDED(attr_set)) {
    const int max_lds = 160 * 1024;
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    RN_ATTRS_DONE(attr_set);
  }
  BneckArgs a;
  a.x = (const uint16_t*)p->x; a.y = (uint16_t*)p->y;
  const uint16_t* w = (const uint16_t*)p->w_packed;
  a.wa = w; w += bneck_frag_elems(1, p->Cx, 64);
  a.wb = w; w += bneck_frag_elems(9, 64, 64);
  a.wo = w; w += bneck_frag_elems(1, 64, 256);
  a.wsc = p->Cx == 64 ? w : nullptr;
  a.affine = p->affine;
  a.N = p->N; a.H = p->H; a.W = p->W; a.Cx = p->Cx;
  // one workgroup per CU (the LDS holds Wb and the ring once): cut every image into row segments until the grid covers the
  // chip; a segment recomputes two halo rows of stage A, so no finer than the chip needs
  int cus = rn_num_cus() - p->opts.reserved_cus;
  if (p->opts.max_workgroups > 0 && p->opts.max_workgroups < cus) cus = p->opts.max_workgroups;
  if (cus < 1) cus = 1;
  int segs = (cus + p->N - 1) / p->N;
  if (segs > p->H) segs = p->H;
  if (segs < 1) segs = 1;
  a.rows_per_wg = (p->H + segs - 1) / segs;
  a.segs = (p->H + a.rows_per_wg - 1) / a.rows_per_wg;
  const dim3 grid((unsigned)(p->N * a.segs)), block((unsigned)(p->W / 32 * 128));   // W / 32 consumer + W / 32 producer waves
  hipStream_t st = (hipStream_t)stream;
#define BNECK_LAUNCH(DBG_)                                                                                   \
  do {                                                                                                       \
    if (p->Cx == 256) {                                                                                      \
      if (slots == 4) hipLaunchKernelGGL((bneck64_kernel<256, 4, DBG_>), grid, block, lds, st, a);          \
      else hipLaunchKernelGGL((bneck64_kernel<256, 3, DBG_>), grid, block, lds, st, a);                      \
    } else {                                                                                                 \
      if (slots == 4) hipLaunchKernelGGL((bneck64_kernel<64, 4, DBG_>), grid, block, lds, st, a);           \
      else hipLaunchKernelGGL((bneck64_kernel<64, 3, DBG_>), grid, block, lds, st, a);                       \
    }                                                                                                        \
  } while (0)
#ifdef RN_PROBES
#define BNECK_ATTR(DBG_)                                                                                                      \
  RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<256, 4, DBG_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<64, 4, DBG_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<256, 3, DBG_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
  RN_CHECK_HIP(hipFuncSetAttribute((const void*)bneck64_kernel<64, 3, DBG_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  switch (p->opts.ablate) {
    case 1: BNECK_ATTR(1) BNECK_LAUNCH(1); break;
    case 2: BNECK_ATTR(2) BNECK_LAUNCH(2); break;
    case 3: BNECK_ATTR(3) BNECK_LAUNCH(3); break;
    case 4: BNECK_ATTR(4) BNECK_LAUNCH(4); break;
    case 8: BNECK_ATTR(8) BNECK_LAUNCH(8); break;
    case 16: BNECK_ATTR(16) BNECK_LAUNCH(16); break;
    case 32: BNECK_ATTR(32) BNECK_LAUNCH(32); break;
    case 64: BNECK_ATTR(64) BNECK_LAUNCH(64); break;
    case 72: BNECK_ATTR(72) BNECK_LAUNCH(72); break;
    case 7: BNECK_ATTR(7) BNECK_LAUNCH(7); break;
    case 127: BNECK_ATTR(127) BNECK_LAUNCH(127); break;
    default: BNECK_LAUNCH(0); break;
  }
#undef BNECK_ATTR
#else
  BNECK_LAUNCH(0);
#endif
#undef BNECK_LAUNCH
  RN_CHECK_LAUNCH();
  return RN_OK;
}
