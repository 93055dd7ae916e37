// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 value = element index.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // each lane supplies the address of row (l & 15)... try: addr = row*stride + (l>>4)*4 elements
  unsigned addr = (unsigned)(uintptr_t)(&lds[(l & 15) * stride_elems + (l >> 4) * 4]);
  s4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {16, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d (lane: supplied elem addr -> 4 values)\n", stride);
    for (int l = 0; l < 64; ++l) printf("l%2d a=%4d : %4d %4d %4d %4d\n", l, (l & 15) * stride + (l >> 4) * 4, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
