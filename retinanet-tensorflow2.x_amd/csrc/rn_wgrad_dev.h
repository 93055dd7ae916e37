// rn_wgrad_dev.h — declarations shared by the weight-gradient kernels (rn_wgrad.hip: 128 x 128 per-tap tiles;
// rn_wgrad_big.hip: 256 x 256 per-tap tiles, ping-pong).
#ifndef RN_WGRAD_DEV_H_
#define RN_WGRAD_DEV_H_
#include "rn_common.h"

typedef rn_h16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef rn_h16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
// ds_read_b64_tr_b16 of four 16-bit elements: the builtin is typed per element type (bf16: ext vector of __bf16;
// f16: GCC vector of __fp16), the register contents are the same 64 bits either way
#ifdef RN_F16
typedef __fp16 rn_tr4_t __attribute__((__vector_size__(8)));
#else
typedef bf16x4_t rn_tr4_t;
#endif
typedef __attribute__((address_space(3))) rn_tr4_t lds_b4_t;
__device__ __forceinline__ bf16x4_t rn_ds_read_tr4(const lds_b4_t* p) {
  return __builtin_bit_cast(bf16x4_t, RN_DS_READ_TR16_B64((lds_b4_t*)p));
}

// ---- transpose reads the COMPILER DOES NOT SEE -------------------------------------------------------------------
// hipcc (ROCm 7.2) treats the ds_read_tr builtin as an LDS read that may alias a pending LDS-DMA (buffer_load ... lds
// is a pending LDS write on the VM counter) and puts `s_waitcnt vmcnt(0)` in front of the first transpose read of every
// K step: the whole DMA prefetch ring is drained once per step and the counted vmcnt(N) waits of the kernels are moot
// (measured on the ISA of all three weight-gradient kernels; the ds_read_b128 loads of the forward kernels do not
// trigger it).  Issued from inline asm the reads are invisible to that pass; the kernels already order LDS-DMA data
// themselves (counted vmcnt, then s_barrier, then the read — MI355X_MICROARCH.md, "Two waves per SIMD" item 7).
// Usage (cdna_hip_programming.md 5.7, form (ii)): rn_tr_issue() for every read, then ONE rn_tr_wait*() that names
// every destination "+v" before the first consumer, then __builtin_amdgcn_sched_barrier(0) in front of the MFMAs.
typedef unsigned rn_u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rn_lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
#define RN_TR_ISSUE(dst_, addr_, off_) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "i"(off_))
__device__ __forceinline__ bf16x8_t rn_tr_frag(rn_u32x2_t lo, rn_u32x2_t hi) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return __builtin_bit_cast(bf16x8_t, v);
}

#define WG_THREADS 256
#define WG_BK 64
#define WG_TILE_BYTES (WG_BK * 256)
#define WG_OOB 0x80000000u

struct WgSegDev {
  const uint16_t* x;
  const uint16_t* dy;
  int N, H, W, Ho, Wo, P, chunk_begin, dyS, xS, pad_;
};

struct WgArgs {
  int R, S, sh, sw, pt, pl, nseg, total_chunks, CH, co_tiles, ci_tiles, Cin, Cout, pad_;
  int co_groups, gco;   // the co tiles are walked in `co_groups` groups of `gco`
  float* ws;
  WgSegDev seg[RN_CONV_MAX_SEGMENTS];
};


// rn_wgrad_halo.hip: all nine taps of a 3x3 / stride 1 / pad 1 layer in one workgroup, reduction over image rows
#define RN_WGRAD_MAX_GROUP 8
struct WhSeg {   // geometry of a segment: shared by every layer of a grouped launch
  int N, H, W, xS, dyS;
  int ctiles;       // column strips of 16 pixels
  int L;            // steps per strip (two padded rows each), the strip's load-only step included
  int step_begin;   // first step of the segment in the launch's step sequence
  int pad_;
};
struct WhPtr {
  const uint16_t* x;
  const uint16_t* dy;
};
struct WhArgs {
  int nseg, Cin, Cout, co_tiles, ci_tiles, total_steps, CHs, total_chunks;
  int ngroups, pad_;      // layers of identical geometry in this launch (rn_conv2d_nhwc_wgrad_group): tiles = groups x co x ci
  float* ws;              // [group][chunk][co][tap][ci]
  WhSeg seg[RN_CONV_MAX_SEGMENTS];
  WhPtr ptr[RN_WGRAD_MAX_GROUP][RN_CONV_MAX_SEGMENTS];
};
struct WgDwPtrs {         // output tensors of the (grouped) split-K reduction
  float4* p[RN_WGRAD_MAX_GROUP];
};
bool rn_wgrad_halo_plan(const rn_wgrad_problem* const* ps, int ngroups, WhArgs& a);
size_t rn_wgrad_halo_workspace_bytes(const WhArgs& a);
int rn_launch_wgrad_halo(const WhArgs& a, const rn_launch_opts& opts, hipStream_t st);

// rn_wgrad_big.hip: eligibility + plan (fills a.CH / chunk_begin / total_chunks / co_tiles / ci_tiles for
// 256-wide tiles) and launch of the partial-tile kernel (same workspace layout as wgrad_kernel)
bool rn_wgrad_big_plan(const rn_wgrad_problem* p, WgArgs& a);
int rn_launch_wgrad_big(const WgArgs& a, const rn_launch_opts& opts, hipStream_t st);
#endif  // RN_WGRAD_DEV_H_
