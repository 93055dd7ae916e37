// Which K-loop organisation feeds the matrix pipe best when the fragments come out of LDS?  (the
// 3x3 kernels sit at MFMA-busy 0.55 - 0.60; per K step the eight 128 x 64 wave tiles of a 256 x 256 workgroup tile read
// 96 KB of fragments, 0.75 fragment per MFMA.)  Synthetic K loops over a four-stage LDS ring that an LDS-DMA stream keeps
// refilling from an L2-resident buffer (32 KB per K step of 32, like the 1x1 kernel), no epilogue, one workgroup per CU:
//   pp8  : 8 waves, wave tile 128 x 64 (4 x 2 MFMA tiles, 128 accumulators), the two waves of a SIMD ping-pong
//          "16 MFMAs from registers" / "12 fragment reads + DMA pieces + counted wait", two barriers per K step
//          (conv_big_kernel / conv_halo_kernel today);
//   db8  : the same wave tiles, no ping-pong: every wave reads the NEXT step's fragments into a second register set
//          between the MFMAs of the current step, ONE barrier per K step;
//   db4  : 4 waves (one per SIMD), wave tile 128 x 128 (4 x 4 MFMA tiles, 256 accumulators), second fragment set, one
//          barrier per K step: 0.5 fragment per MFMA, 64 KB of fragment reads per K step instead of 96.
// hipcc --offload-arch=gfx950 -O3 wave_tile_probe.hip -o wave_tile_probe && ./wave_tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int STAGE = 32 * 1024;   // 256 rows x 64 B of "pixels" + 256 rows x 64 B of "weights"
constexpr int STAGES = 4;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}
// fragment of 32-row tile `tile` of a 16 KB operand image, K slice kk: row = lane & 31, 16-byte slot 2 kk + lane / 32,
// XOR-swizzled by (row / 4) & 3 (conflict free for ds_read_b128, the layout of the conv kernels)
__device__ __forceinline__ int frag_off(int lane) {
  const int row = lane & 31, h = lane >> 5;
  return row * 64 + ((h ^ ((row >> 2) & 3)) << 4);
}

template <int MODE>   // 0 pp8, 1 db8, 2 db4
__global__ void __launch_bounds__(MODE == 2 ? 256 : 512) k(const void* src, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = MODE == 2 ? 4 : 8, TI = 4, TJ = MODE == 2 ? 4 : 2;
  constexpr int PIECES = 32 / NW;   // 1 KB DMA pieces per wave and K step
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = MODE == 2 ? wave >> 1 : wave >> 2, wave_n = MODE == 2 ? wave & 1 : wave & 3;
  const int off0 = frag_off(lane);
  const int offA = wave_m * (TI * 2048) + off0, offB = 16384 + wave_n * (TJ * 2048) + off0;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
  f32x16_t acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  unsigned g_src = (unsigned)(blockIdx.x * 4096 + lane * 16);
#define ISSUE(step_)                                                                                  \
  do {                                                                                                \
    char* st__ = smem + ((step_) & (STAGES - 1)) * STAGE;                                             \
    _Pragma("unroll") for (int q = 0; q < PIECES; ++q)                                                \
      dma16(rs, st__ + (q * NW + wave) * 1024, (g_src + (unsigned)((step_) * 32768 + (q * NW + wave) * 1024)) & 0xfffffu); \
  } while (0)
#define READ(set_, step_)                                                                             \
  do {                                                                                                \
    const char* st__ = smem + ((step_) & (STAGES - 1)) * STAGE;                                       \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                \
      _Pragma("unroll") for (int j = 0; j < TJ; ++j) fb[set_][kk][j] = *(const bf16x8_t*)(st__ + (offB ^ (kk << 5)) + j * 2048); \
      _Pragma("unroll") for (int i = 0; i < TI; ++i) fa[set_][kk][i] = *(const bf16x8_t*)(st__ + (offA ^ (kk << 5)) + i * 2048); \
    }                                                                                                 \
  } while (0)
#define MFMAS(set_)                                                                                   \
  do {                                                                                                \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                  \
      _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                  \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set_][kk][j], fa[set_][kk][i], acc[i][j], 0, 0, 0); \
  } while (0)
  bf16x8_t fa[2][2][TI], fb[2][2][TJ];
  // prologue: stages 0..2 in flight, 0 and 1 landed
  ISSUE(0); ISSUE(1); ISSUE(2);
  if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (MODE == 0) {
    const int grp = wave >> 2;
    READ(0, 0);
    ISSUE(3);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // steps 0, 1 landed
    if (grp == 1) __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int g = 0; g < steps; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      MFMAS(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      READ(0, g + 1);
      ISSUE(g + 4);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // everything but the last two steps' pieces
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
    READ(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int g = 0; g < steps; g += 2) {
      // step g on set 0 while set 1 fills with step g + 1; then the other way round
      READ(1, g + 1);
      ISSUE(g + 3);
      MFMAS(0);
      if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      READ(0, g + 2);
      ISSUE(g + 4);
      MFMAS(1);
      if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  out[blockIdx.x * blockDim.x + tid] = t;
}

int main() {
  const int blocks = 256, steps = 4096;
  void* src; float* out;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, blocks * 512 * 4);
  uint16_t* h = (uint16_t*)malloc(1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = STAGES * STAGE;
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int zero = 0; zero < 2; ++zero) {
    for (int i = 0; i < (1 << 19); ++i) {
      const float f = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
      uint32_t u; memcpy(&u, &f, 4);
      h[i] = (uint16_t)(u >> 16);
    }
    hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), lds, 0, src, out, steps);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), lds, 0, src, out, steps);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), lds, 0, src, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double flops = (double)blocks * steps * 2.0 * 256 * 256 * 32;
      printf("%s %s: %.3f ms  %.1f TFLOP/s  (err %d)\n", zero ? "zeros " : "random", mode == 0 ? "pp8" : (mode == 1 ? "db8" : "db4"),
             best, flops / best / 1e9, (int)hipGetLastError());
    }
  }
  return 0;
}
