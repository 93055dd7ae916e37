// What a 2-read + 1-write elementwise pass (the shape of bn_bwd_apply_kernel) can reach on this chip, by launch form:
// rows per thread iteration (loads in flight), non-temporal loads / stores, workgroups, threads per workgroup.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_probe.hip -o tools/probes/bin/stream_probe && tools/probes/bin/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld(const uint4* p, bool nt) {
  u32x4 v = nt ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st(uint4* p, uint4 r, bool nt) {
  u32x4 v = {r.x, r.y, r.z, r.w};
  if (nt) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
}

template <int U, bool NTL, bool NTS, int NREAD>
__global__ void __launch_bounds__(512) k_stream(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o,
                                                long long total) {
  const long long lanes = (long long)gridDim.x * blockDim.x;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * lanes < total; i += U * lanes) {
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = ld(a + i + u * lanes, NTL);
      if (NREAD > 1) y[u] = ld(b + i + u * lanes, NTL);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint4 r;
      if (NREAD > 1) {
        r.x = x[u].x ^ y[u].x; r.y = x[u].y + y[u].y; r.z = x[u].z ^ y[u].z; r.w = x[u].w + y[u].w;
      } else {
        r.x = x[u].x * 3u; r.y = x[u].y + 1u; r.z = x[u].z ^ 5u; r.w = x[u].w + 7u;
      }
      st(o + i + u * lanes, r, NTS);
    }
  }
  for (; i < total; i += lanes) {
    uint4 r = a[i];
    if (NREAD > 1) { const uint4 y = b[i]; r.x ^= y.x; r.y += y.y; r.z ^= y.z; r.w += y.w; }
    o[i] = r;
  }
}

// torch-like: no grid-stride loop — block b owns the contiguous chunk [b * T * U, (b + 1) * T * U) of 16-byte units, every thread
// issues its U loads (T units apart) before the first use
template <int U, int NREAD>
__global__ void __launch_bounds__(256) k_chunk(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o,
                                               long long total) {
  const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
  uint4 x[U], y[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = base + u * 256;
    if (i < total) { x[u] = a[i]; if (NREAD > 1) y[u] = b[i]; }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = base + u * 256;
    if (i < total) {
      uint4 r;
      if (NREAD > 1) { r.x = x[u].x ^ y[u].x; r.y = x[u].y + y[u].y; r.z = x[u].z ^ y[u].z; r.w = x[u].w + y[u].w; }
      else { r.x = x[u].x * 3u; r.y = x[u].y + 1u; r.z = x[u].z ^ 5u; r.w = x[u].w + 7u; }
      o[i] = r;
    }
  }
}
// grid-stride over such chunks (persistent workgroups)
template <int U, int NREAD>
__global__ void __launch_bounds__(256) k_chunk_loop(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o,
                                                    long long total) {
  for (long long c = blockIdx.x; c * (256 * U) < total; c += gridDim.x) {
    const long long base = c * (256 * U) + threadIdx.x;
    uint4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = base + u * 256;
      if (i < total) { x[u] = a[i]; if (NREAD > 1) y[u] = b[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = base + u * 256;
      if (i < total) {
        uint4 r;
        if (NREAD > 1) { r.x = x[u].x ^ y[u].x; r.y = x[u].y + y[u].y; r.z = x[u].z ^ y[u].z; r.w = x[u].w + y[u].w; }
        else { r.x = x[u].x * 3u; r.y = x[u].y + 1u; r.z = x[u].z ^ 5u; r.w = x[u].w + 7u; }
        o[i] = r;
      }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool NTL, bool NTS, int NREAD>
static void run(const char* name, const uint4* a, const uint4* b, uint4* o, long long total, int blocks, int threads) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<U, NTL, NTS, NREAD>), dim3(blocks), dim3(threads), 0, 0, a, b, o, total);
  CK(hipEventRecord(e0, 0));
  const int it = 10;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k_stream<U, NTL, NTS, NREAD>), dim3(blocks), dim3(threads), 0, 0, a, b, o, total);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= it;
  const double bytes = (double)total * 16 * (NREAD + 1);
  printf("  %-28s blocks %5d x %3d  %8.1f us  %6.2f TB/s\n", name, blocks, threads, ms * 1e3, bytes / ms / 1e9);
}

template <int U, int NREAD, bool LOOP>
static void run_chunk(const char* name, const uint4* a, const uint4* b, uint4* o, long long total, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = LOOP ? blocks : (int)((total + 256 * U - 1) / (256 * U));
  auto launch = [&]() {
    if (LOOP) hipLaunchKernelGGL((k_chunk_loop<U, NREAD>), dim3(grid), dim3(256), 0, 0, a, b, o, total);
    else hipLaunchKernelGGL((k_chunk<U, NREAD>), dim3(grid), dim3(256), 0, 0, a, b, o, total);
  };
  for (int w = 0; w < 2; ++w) launch();
  CK(hipEventRecord(e0, 0));
  const int it = 10;
  for (int w = 0; w < it; ++w) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= it;
  const double bytes = (double)total * 16 * (NREAD + 1);
  printf("  %-28s blocks %7d x 256  %8.1f us  %6.2f TB/s\n", name, grid, ms * 1e3, bytes / ms / 1e9);
}

int main() {
  const long long sizes[] = {32LL * 160 * 160 * 256 / 8, 32LL * 80 * 80 * 512 / 8, 32LL * 160 * 160 * 64 / 8};
  const long long mx = sizes[0];
  uint4 *a, *b, *o;
  CK(hipMalloc(&a, mx * 16)); CK(hipMalloc(&b, mx * 16)); CK(hipMalloc(&o, mx * 16));
  CK(hipMemset(a, 1, mx * 16)); CK(hipMemset(b, 2, mx * 16));
  for (long long total : sizes) {
    printf("tensor %.1f MB (2 reads + 1 write)\n", total * 16 / 1e6);
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
      run<1, false, false, 2>("U1", a, b, o, total, blocks, 256);
    }
    run<1, false, false, 2>("U1 512thr", a, b, o, total, 2048, 512);
    run<2, false, false, 2>("U2", a, b, o, total, 4096, 256);
    run<2, false, false, 2>("U2", a, b, o, total, 2048, 256);
    run<4, false, false, 2>("U4", a, b, o, total, 2048, 256);
    run<4, false, false, 2>("U4", a, b, o, total, 1024, 256);
    run<1, true, false, 2>("U1 nt-load", a, b, o, total, 4096, 256);
    run<1, false, true, 2>("U1 nt-store", a, b, o, total, 4096, 256);
    run<1, true, true, 2>("U1 nt-both", a, b, o, total, 4096, 256);
    run<2, true, true, 2>("U2 nt-both", a, b, o, total, 4096, 256);
    run<4, true, true, 2>("U4 nt-both", a, b, o, total, 2048, 256);
    run_chunk<1, 2, false>("chunk U1 (one block per 4 KB)", a, b, o, total, 0);
    run_chunk<2, 2, false>("chunk U2", a, b, o, total, 0);
    run_chunk<4, 2, false>("chunk U4 (torch-like)", a, b, o, total, 0);
    run_chunk<8, 2, false>("chunk U8", a, b, o, total, 0);
    run_chunk<4, 2, true>("chunk U4 loop", a, b, o, total, 2048);
    run_chunk<4, 2, true>("chunk U4 loop", a, b, o, total, 4096);
    run_chunk<8, 2, true>("chunk U8 loop", a, b, o, total, 2048);
    printf("tensor %.1f MB (1 read + 1 write)\n", total * 16 / 1e6);
    run_chunk<4, 1, false>("chunk U4 (torch-like)", a, b, o, total, 0);
    run_chunk<4, 1, true>("chunk U4 loop", a, b, o, total, 2048);
    run<1, false, false, 1>("U1", a, b, o, total, 4096, 256);
    run<2, false, false, 1>("U2", a, b, o, total, 4096, 256);
    run<1, true, true, 1>("U1 nt-both", a, b, o, total, 4096, 256);
    run<4, true, true, 1>("U4 nt-both", a, b, o, total, 2048, 256);
  }
  return 0;
}
