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


// rn_wgrad_big.hip: eligibility + plan (fills a.CH / chunk_begin / total_chunks / co_tiles / ci_tiles for
// 256-wide tiles) and launch of the partial-tile kernel (same workspace layout as wgrad_kernel)
bool rn_wgrad_big_plan(const rn_wgrad_problem* p, WgArgs& a);
int rn_launch_wgrad_big(const WgArgs& a, const rn_launch_opts& opts, hipStream_t st);
#endif  // RN_WGRAD_DEV_H_
