// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate on gfx950 (no memory traffic), with and
// without an s_barrier every 16 MFMAs, 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int BARRIER>
__global__ void __launch_bounds__(512) probe(const float* in, float* out, int iters) {
  bf16x8_t a[4], b[2];
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 8; ++q) a[i][q] = (__bf16)in[(threadIdx.x * 8 + q + i * 7) & 1023];
  for (int i = 0; i < 2; ++i)
    for (int q = 0; q < 8; ++q) b[i][q] = (__bf16)in[(threadIdx.x * 8 + q + i * 13 + 5) & 1023];
  f32x16_t acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wave = threadIdx.x >> 6;
  if (BARRIER == 2 && wave >= 4) __builtin_amdgcn_s_barrier();   // phase shift: ping-pong
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    if (BARRIER == 1) __builtin_amdgcn_s_barrier();
    if (BARRIER == 2) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
  }
  if (BARRIER == 2 && wave < 4) __builtin_amdgcn_s_barrier();
  float t = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main(int argc, char** argv) {
  const int iters = 4096, blocks = 256 * 4;
  float *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, blocks * 512 * 4);
  float h[1024];
  for (int zero = 0; zero < 2; ++zero) {
    for (int i = 0; i < 1024; ++i) h[i] = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode)
      for (int threads = 256; threads <= 512; threads += 256) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
          if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
          if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (threads / 64) * iters * 16 * 32768.0;
        printf("%s barrier_mode=%d threads=%d: %.3f ms  %.1f TFLOP/s\n", zero ? "zeros " : "random", mode, threads, ms,
               flops / ms / 1e9);
      }
  }
  return 0;
}
