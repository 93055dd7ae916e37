// rn_probe.hip — measurement infrastructure for the roofline of bench.py: what the matrix pipe of THIS
// part sustains, measured live, beside the nominal peak.
//
// rn_probe_mfma runs nothing but v_mfma_f32_32x32x16 (the instruction of every implicit-GEMM kernel here: 8 waves per
// workgroup = 2 per SIMD, 4 workgroups per CU, accumulator-chained like a wave tile of 4 x 2 MFMA tiles), no memory
// traffic, operands taken from a 1024-float table the caller fills — random values or zeros.  The chip clocks to its
// power budget (MI355X_MICROARCH.md, "DVFS give-back"): the same instruction stream runs ~2.4 GHz on all-zero
// operands and markedly lower on random ones, so the nominal 2.5 PFLOP/s (2.4 GHz) is not reachable on real data.
// Workgroup 0 stamps the shader clock (s_memtime) and the 100 MHz wall clock at its first and last instruction: the
// ratio is the core clock the launch actually ran at.
#include "rn_common.h"

typedef rn_h16 pr_h16x8_t __attribute__((ext_vector_type(8)));
typedef float pr_f32x16_t __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) probe_mfma_kernel(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                         unsigned long long* __restrict__ clocks) {
  unsigned long long c0 = 0, w0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    c0 = __builtin_readcyclecounter();   // s_memtime: shader-clock ticks
    w0 = wall_clock64();                 // 100 MHz
  }
  pr_h16x8_t a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 8; ++q) a[i][q] = (rn_h16)in[(threadIdx.x * 8 + q + i * 7) & 1023];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 8; ++q) b[i][q] = (rn_h16)in[(threadIdx.x * 8 + q + i * 13 + 5) & 1023];
  pr_f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = RN_MFMA_32x32x16(b[j], a[i], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clocks[0] = c0;
    clocks[1] = __builtin_readcyclecounter();
    clocks[2] = w0;
    clocks[3] = wall_clock64();
  }
}

extern "C" long long rn_probe_mfma_flops(int iters) {   // FLOPs one rn_probe_mfma launch executes
  return iters > 0 ? (long long)rn_num_cus() * 4 * 8 * iters * 16 * 32768ll : 0;
}

extern "C" int rn_probe_mfma(const float* table, float* out, int iters, unsigned long long* clocks, void* stream) {
  RN_CHECK_ARG(table && out && clocks && iters > 0, "rn_probe_mfma: bad argument");
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(rn_num_cus() * 4), dim3(512), 0, (hipStream_t)stream, table, out, iters, clocks);
  RN_CHECK_LAUNCH();
  return RN_OK;
}

// ---- do two streams really run side by side? ---------------------------------------------------------------------
// HIP maps every stream onto one of a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) by creation order; two streams
// that share a queue execute strictly in submission order, and a "wait for that event" packet of one of them then blocks the
// other's kernels too.  Measured (round 5, profiles/r05_ab): the data-parallel step's third stream landed on the weight-
// gradient stream's queue in one process (+3 ms per step) and on a worse one inside bench.py (+6.5 ms) — which streams
// alias depends on how many streams the process created before.  rn_probe_spin is the probe the engines pick their side
// streams with (retinanet/_C.py::concurrent_stream): one wave that does nothing for `microseconds`; launched on stream A,
// a second, short one on stream B finishes first only if A and B are on different queues.
__global__ void probe_spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();   // 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int rn_probe_spin(int microseconds, void* stream) {
  RN_CHECK_ARG(microseconds >= 0 && microseconds <= 100000, "rn_probe_spin: %d us (0..100000)", microseconds);
  hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
