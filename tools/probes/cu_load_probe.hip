// Per-CU operand-staging bandwidth from L2: what one workgroup (4 or 8 waves) can pull into LDS per microsecond,
//   (a) LDS-DMA:  buffer_load_dwordx4 ... lds (1 KiB per wave instruction, the conv kernels' path)
//   (b) VGPR path: buffer_load_dwordx4 into registers, then ds_write_b128
// with K pieces in flight per wave.  One workgroup per CU; the source is a 2 MB region every workgroup re-reads (L2 /
// MALL resident), addressed like an im2col A tile (8 lanes per 128-byte row segment, rows 4 KB apart).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/cu_load_probe.hip -o tools/probes/bin/cu_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int PIECES>   // MODE 0: LDS-DMA, 1: VGPR + ds_write
__global__ void __launch_bounds__(512) k_load(const char* src, unsigned src_bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
  // lane -> (row = lane / 8, 16-byte chunk = lane % 8); rows 4 KB apart (a 2048-channel pixel pitch)
  const unsigned lane_off = (unsigned)((lane >> 3) * 4096 + (lane & 7) * 16);
  const unsigned region = 2u << 20;   // every workgroup walks the same 2 MB (+ a row span): L2 resident
  unsigned base = (unsigned)(blockIdx.x * 65536 + wave * 32768) % region;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        char* dst = smem + ((wave * PIECES + p) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, (int)(base + lane_off + p * 128), 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[PIECES];
#pragma unroll
      for (int p = 0; p < PIECES; ++p) v[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + lane_off + p * 128), 0, 0);
#pragma unroll
      for (int p = 0; p < PIECES; ++p) *(u32x4*)(smem + ((wave * PIECES + p) * 1024) + lane * 16) = v[p];
    }
    base += 1024 * 37;
    if (base >= region) base -= region;
  }
  __syncthreads();
  acc = *(unsigned*)(smem + threadIdx.x * 4);
  if (acc == 0x12345678u) sink[0] = acc;
  (void)nw;
}

template <int MODE, int PIECES>
static void run(const char* name, const char* src, unsigned bytes, unsigned* sink, int grid, int threads) {
  const int iters = 2000;
  const int lds = (threads / 64) * PIECES * 1024;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_load<MODE, PIECES>), dim3(grid), dim3(threads), lds, 0, src, bytes, 100, sink);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_load<MODE, PIECES>), dim3(grid), dim3(threads), lds, 0, src, bytes, iters, sink);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double per_wg = (double)iters * (threads / 64) * PIECES * 1024;
  printf("  %-10s pieces/wave %2d  grid %3d x %3d thr: %7.1f GB/s per workgroup, %6.2f TB/s total\n", name, PIECES, grid, threads,
         per_wg / ms / 1e6, per_wg * grid / ms / 1e9);
}

int main() {
  const unsigned bytes = 32u << 20;
  char* src; unsigned* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(src, 1, bytes));
  for (int grid : {64, 128, 256}) {
    for (int threads : {256, 512}) {
      run<0, 4>("lds-dma", src, bytes, sink, grid, threads);
      run<0, 8>("lds-dma", src, bytes, sink, grid, threads);
      run<0, 16>("lds-dma", src, bytes, sink, grid, threads);
      run<1, 4>("vgpr", src, bytes, sink, grid, threads);
      run<1, 8>("vgpr", src, bytes, sink, grid, threads);
      run<1, 16>("vgpr", src, bytes, sink, grid, threads);
    }
  }
  return 0;
}
