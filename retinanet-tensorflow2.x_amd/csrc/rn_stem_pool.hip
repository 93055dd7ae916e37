// rn_stem_pool.hip — ResNet stem in one kernel: Conv2D 7x7 / stride 2 (fixed padding 3) + BatchNorm affine (inference
// form: the reference freezes `resnet_initial`, builder.py:28-29, and serving always folds) + ReLU + MaxPool 3x3 / 2 SAME
// (resnet.py:288-307).  The separate launches move the 64-channel stem output (32 x 320 x 320 x 64 bf16 = 419 MB at
// the bench sizes) to HBM and back, and the 7 x 1 x 32 implicit GEMM re-stages every input pixel for each of the seven
// filter rows through the LDS-DMA path (244 TFLOP/s, 385 us); here a workgroup
//   * stages the bf16 NHWC4 input patch of its tile ONCE (15 rows x 136 pixels x 8 B = 16 KB for 2 x 32 pooled pixels),
//     and reads MFMA fragments straight out of it: the window of conv pixel (cr, cc) for filter row r is the 8
//     consecutive pixels 2cc .. 2cc+7 of patch row 2cr + r, i.e. 64 contiguous bytes at a 16-byte aligned address —
//     neighbouring conv pixels' windows overlap in LDS instead of being duplicated by an im2col;
//   * keeps the whole filter (64 x 224 bf16 = 28 KB, K-step major) in LDS for all its tiles (persistent walk);
//   * writes the 5 x 65 conv pixels its pooled tile needs (halo of one row / column recomputed: +27 % of a GEMM that
//     is ~45 us of MFMA time in total) as bf16 into LDS and pools from there: only the pooled tensor goes to HBM.
// Arithmetic = the unfused launches': fp32 accumulation in the same K order (r major, 2 x 16 per row), acc -> bf16
// (the Conv2D output tensor) -> *scale + shift -> relu -> bf16, max over the valid taps.  Masked conv pixels are
// written as 0: every valid value is >= 0 after the ReLU and every pooling window holds a valid pixel, so the max over
// the window with zeros equals the max over its valid taps (the unfused kernel skips them).
// HBM-bound in principle (107 MB in, 105 MB out at the bench sizes).
#include "rn_conv_dev.h"

namespace {

constexpr int PH = 2, PW = 32;                 // pooled tile
constexpr int CR = 2 * PH + 1, CC = 2 * PW + 1;   // conv pixels it needs: 5 x 65
constexpr int CPIX = CR * CC;                  // 325
constexpr int MBLK = (CPIX + 31) / 32;         // 11 MFMA row blocks
constexpr int PR = 2 * CR + 5, PC = 2 * CC + 6;   // input patch: 15 rows x 136 pixels
constexpr int PROW_BYTES = PC * 8;             // 1088
constexpr int PCHUNKS = PR * (PROW_BYTES / 16);   // 1020 16-byte chunks
constexpr int KSTEPS = 14;                     // 7 filter rows x 2 K slices of 16
constexpr int W_BYTES = KSTEPS * 64 * 32;      // filter, [step][cout][16] bf16: 28 KB
constexpr int CT_BYTES = MBLK * 32 * 128;      // conv tile [352 px][64 ch] bf16: 44 KB (the patch aliases its head)
constexpr int AFF_OFF = W_BYTES + CT_BYTES;    // scale | shift, f32[64] each
constexpr int LDS_BYTES = AFF_OFF + 512;
constexpr int NTHREADS = 256;

struct StemPoolArgs {
  const uint16_t* x;      // bf16 [N][Hp][Wp][4]
  const uint16_t* w;      // bf16 [64][7][32]
  const float* scale;
  const float* shift;
  uint16_t* y;            // bf16 [N][Po][Qo][64]
  int N, Hp, Wp, Hs, Ws, Po, Qo, pt, pl, tiles_x, tiles_y, total_tiles, relu6;
};

typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sp_pk_max(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, a), __builtin_bit_cast(i16x2_t, b)));
}
__device__ __forceinline__ uint32_t sp_pack2(float lo, float hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef rn_h16 b2 __attribute__((ext_vector_type(2)));
  f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

__global__ void __launch_bounds__(NTHREADS, 2) stem_pool_kernel(const StemPoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem;                 // filter
  char* const ct = smem + W_BYTES;       // conv tile; the input patch lives in its first 16 KB until the MFMAs are done
  float* const aff = (float*)(smem + AFF_OFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 31, fh = lane >> 5;

  // filter -> LDS, K-step major: the fragment of step t is 2 KB contiguous (lane n + 32*half reads 16 bytes)
  for (int i = tid; i < KSTEPS * 64 * 2; i += NTHREADS) {
    const int half = i & 1, n = (i >> 1) & 63, t = i >> 7;
    *(uint4*)(wl + (t * 64 + n) * 32 + half * 16) = *(const uint4*)(a.w + n * 224 + t * 16 + half * 8);
  }
  if (tid < 64) { aff[tid] = a.scale ? a.scale[tid] : 1.0f; aff[64 + tid] = a.shift ? a.shift[tid] : 0.0f; }

  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long long)a.N * a.Hp * a.Wp * 8), 0x00020000);

#pragma unroll 1
  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x;
    const int t2 = tile / a.tiles_x;
    const int ty = t2 % a.tiles_y, n = t2 / a.tiles_y;
    const int py0 = ty * PH, px0 = tx * PW;
    const int cr0 = 2 * py0 - a.pt, cc0 = 2 * px0 - a.pl;     // first conv pixel of the tile (may be -1)
    __syncthreads();   // the previous tile's pooling reads of `ct` (and, first tile, the filter fill) are done

    // ---- input patch: rows 2*cr0 .. +14, pixels 2*cc0 .. +135, 16-byte chunks, lane-linear in LDS ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = (j * 4 + wave) * 64 + lane;
      const int row = q / (PROW_BYTES / 16), cj = q - row * (PROW_BYTES / 16);
      const int prow = 2 * cr0 + row, pcol = 2 * cc0 + 2 * cj;
      const bool ok = q < PCHUNKS && (unsigned)prow < (unsigned)a.Hp && (unsigned)pcol < (unsigned)a.Wp;
      const unsigned off = ok ? (unsigned)((((long long)n * a.Hp + prow) * a.Wp + pcol) * 8) : RN_OOB;
      dma16(rs_x, ct + (j * 4 + wave) * 1024, off);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- GEMM: wave w owns row blocks w, w+4, w+8 (32 conv pixels each) x 64 channels ----
    f32x16_t acc[3][2];
    int base[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][j][q] = 0.0f;
      int m = (wave + 4 * b) * 32 + fr;
      m = m < CPIX ? m : CPIX - 1;
      const int crl = m / CC, ccl = m - crl * CC;
      base[b] = (2 * crl * PC + 2 * ccl) * 8 + fh * 16;
    }
    const bool last_live = wave + 8 < MBLK;   // block 11 does not exist (wave 3)
#pragma unroll
    for (int t = 0; t < KSTEPS; ++t) {
      const int r = t >> 1, half = t & 1;
      bf16x8_t wf[2], pf[3];
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[j] = *(const bf16x8_t*)(wl + (t * 64 + j * 32 + fr) * 32 + fh * 16);
#pragma unroll
      for (int b = 0; b < 3; ++b) pf[b] = *(const bf16x8_t*)(ct + base[b] + r * PROW_BYTES + half * 32);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        if (b == 2 && !last_live) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[b][j] = RN_MFMA_32x32x16(wf[j], pf[b], acc[b][j], 0, 0, 0);
      }
    }
    __syncthreads();   // every wave is done reading the patch: the conv tile may overwrite it

    // ---- epilogue: acc[b][j][4g + e] = pixel (block row fr), channel j*32 + 8g + 4*fh + e ----
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      if (b == 2 && !last_live) continue;
      const int m = (wave + 4 * b) * 32 + fr;
      const int crl = m / CC, ccl = m - crl * CC;
      const int cr = cr0 + crl, cc = cc0 + ccl;
      const bool valid = m < CPIX && (unsigned)cr < (unsigned)a.Hs && (unsigned)cc < (unsigned)a.Ws;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = j * 32 + g * 8 + fh * 4;
          const float4 sc = *(const float4*)(aff + c), sh = *(const float4*)(aff + 64 + c);
          float v[4] = {acc[b][j][g * 4 + 0], acc[b][j][g * 4 + 1], acc[b][j][g * 4 + 2], acc[b][j][g * 4 + 3]};
          const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = rn_rb(v[e]) * s4[e] + h4[e];
            u = u > 0.0f ? u : 0.0f;
            if (a.relu6) u = u < 6.0f ? u : 6.0f;
            v[e] = valid ? u : 0.0f;
          }
          uint2 pk;
          pk.x = sp_pack2(v[0], v[1]);
          pk.y = sp_pack2(v[2], v[3]);
          // [pixel][64 ch], 16-byte units XOR-swizzled by the pixel, the 8-byte half flipped for pixels 8-15 / 24-31
          // of a block (ds_write_b64 is serviced in 16-lane groups: rn_conv_big_epi.h)
          *(uint2*)(ct + m * 128 + (((c >> 3) ^ (m & 7)) << 4) + ((((c >> 2) ^ (m >> 3)) & 1) << 3)) = pk;
        }
    }
    __syncthreads();

    // ---- pool: 64 pooled pixels x 8 channel units, two items per thread ----
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = it * NTHREADS + tid;
      const int u = item & 7, pp = item >> 3;
      const int ppy = pp / PW, ppx = pp - ppy * PW;
      const int py = py0 + ppy, px = px0 + ppx;
      uint4 best = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int m = (2 * ppy + r) * CC + 2 * ppx + s;
          const uint4 t4 = *(const uint4*)(ct + m * 128 + ((u ^ (m & 7)) << 4));
          const uint4 v4 = ((m >> 3) & 1) ? make_uint4(t4.z, t4.w, t4.x, t4.y) : t4;
          best.x = sp_pk_max(best.x, v4.x); best.y = sp_pk_max(best.y, v4.y);
          best.z = sp_pk_max(best.z, v4.z); best.w = sp_pk_max(best.w, v4.w);
        }
      if (py < a.Po && px < a.Qo)
        *(uint4*)(a.y + ((((long long)n * a.Po + py) * a.Qo + px) * 64 + u * 8)) = best;
    }
  }
}

}  // namespace

// x: bf16 [N, Hp, Wp, 4] from rn_pack_image_nhwc4 (pads in front materialised, Wp % 8 == 0); w: rn_pack_stem_weight_rs
// output for R = S = 7, Cout = 64; conv output Hs x Ws (never written), pooled output y bf16 [N, Po, Qo, 64].
extern "C" int rn_stem_conv_bn_relu_pool(const void* x, const void* w_packed, const float* scale, const float* shift,
                                         void* y, int N, int Hp, int Wp, int Hs, int Ws, int R, int Cout, int act,
                                         int pool_k, int pool_stride, int pool_pad_top, int pool_pad_left, int Po,
                                         int Qo, void* stream) {
  RN_CHECK_ARG(x && w_packed && y && N > 0, "rn_stem_conv_bn_relu_pool: null tensor");
  RN_CHECK_ARG(R == 7 && Cout == 64 && pool_k == 3 && pool_stride == 2 && (act == RN_ACT_RELU || act == RN_ACT_RELU6),
               "rn_stem_conv_bn_relu_pool: only the ResNet stem (7x7/2, 64 channels, relu, 3x3/2 pool)");
  RN_CHECK_ARG(pool_pad_top >= 0 && pool_pad_top <= 1 && pool_pad_left >= 0 && pool_pad_left <= 1,
               "rn_stem_conv_bn_relu_pool: pool pads 0 or 1");
  RN_CHECK_ARG(Wp % 8 == 0 && 2 * (Hs - 1) + 7 <= Hp && 2 * (Ws - 1) + 8 <= Wp && (long long)N * Hp * Wp * 8 < (1ll << 31),
               "rn_stem_conv_bn_relu_pool: packed input %d x %d does not cover the %d x %d conv output", Hp, Wp, Hs, Ws);
  // every pooling window must hold one valid conv pixel (the masked ones are zeros)
  RN_CHECK_ARG(2 * (Po - 1) - pool_pad_top < Hs && 2 * (Qo - 1) - pool_pad_left < Ws && Po > 0 && Qo > 0,
               "rn_stem_conv_bn_relu_pool: pooled size %d x %d against conv output %d x %d", Po, Qo, Hs, Ws);
  RN_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) % 16 == 0, "rn_stem_conv_bn_relu_pool: alignment");
  static unsigned long long attr_set = 0;   // one bit per device
  if (RN_ATTRS_NEEDED(attr_set)) {
    RN_CHECK_HIP(hipFuncSetAttribute((const void*)stem_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RN_ATTRS_DONE(attr_set);
  }
  const int num_cu = rn_num_cus();
  StemPoolArgs a;
  a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_packed; a.scale = scale; a.shift = shift; a.y = (uint16_t*)y;
  a.N = N; a.Hp = Hp; a.Wp = Wp; a.Hs = Hs; a.Ws = Ws; a.Po = Po; a.Qo = Qo; a.pt = pool_pad_top; a.pl = pool_pad_left;
  a.tiles_x = (Qo + PW - 1) / PW;
  a.tiles_y = (Po + PH - 1) / PH;
  a.total_tiles = N * a.tiles_x * a.tiles_y;
  a.relu6 = act == RN_ACT_RELU6 ? 1 : 0;
  const int grid = a.total_tiles < 2 * num_cu ? a.total_tiles : 2 * num_cu;   // two workgroups per CU (73 KB of LDS each)
  hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(NTHREADS), LDS_BYTES, (hipStream_t)stream, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
