// rn_common.h — shared host/device helpers for librnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/rnet_hip.h"

void rn_set_error(const char* fmt, ...);
// rn_core.hip: min(items, CUs not reserved for RCCL, opts.max_workgroups); CU count of the current device
int rn_persistent_grid(int work_items, int num_cu, const rn_launch_opts& opts);
int rn_num_cus();
// index of the current device, 0..63 (rn_core.hip).  Kernel function attributes (hipFuncSetAttribute) are PER DEVICE: a
// process-wide `static bool attr_set` left the second GPU of a process without its dynamic-LDS limit.
int rn_device_slot();
// Per-device "kernel attributes are set" flag words.  RN_ATTRS_NEEDED: the bit of the current device is still clear;
// RN_ATTRS_DONE marks it (atomically) AFTER every hipFuncSetAttribute call succeeded — a failed call returns early through
// RN_CHECK_HIP and leaves the bit clear, so the next launch retries instead of failing with an opaque launch error, and a
// second host thread that gets here before the first one is done sets the (idempotent) attributes itself.
#define RN_ATTRS_NEEDED(mask_) (!((__atomic_load_n(&(mask_), __ATOMIC_ACQUIRE) >> rn_device_slot()) & 1ull))
#define RN_ATTRS_DONE(mask_) ((void)__atomic_fetch_or(&(mask_), 1ull << rn_device_slot(), __ATOMIC_RELEASE))
int rn_validate_launch_opts(const rn_launch_opts& opts, const char* who);

#define RN_CHECK_ARG(cond, ...)  \
  do {                           \
    if (!(cond)) {               \
      rn_set_error(__VA_ARGS__); \
      return RN_EINVAL;          \
    }                            \
  } while (0)

#define RN_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      rn_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return RN_EHIP;                                                                   \
    }                                                                                   \
  } while (0)

#define RN_CHECK_LAUNCH() RN_CHECK_HIP(hipGetLastError())

static inline int64_t rn_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t rn_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- the 16-bit storage type of activations and packed weights -------------------------------------------------
// The library is built twice from the same sources: librnet_hip.so with bfloat16 storage (the `mixed_bfloat16`
// policy of the reference's TPU / bf16 configs, __main__.py:76-77) and, with -DRN_F16, librnet_hip_f16.so with IEEE
// half storage (`mixed_float16`, BASELINE config 5: EfficientNet-B3 fp16 mixed precision + LossScaleOptimizer) on
// v_mfma_f32_32x32x16_f16.  Everything below keeps its historical "bf16" name; under RN_F16 it means "the 16-bit type".
#ifdef RN_F16
typedef _Float16 rn_h16;
#define RN_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define RN_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4f16
#define RN_SIX_X2 0x46004600u   /* 6.0 twice */
__device__ __forceinline__ float rn_bf16_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t rn_f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // RNE
__device__ __forceinline__ float rn_lo16(uint32_t u) { return rn_bf16_to_f32((uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float rn_hi16(uint32_t u) { return rn_bf16_to_f32((uint16_t)(u >> 16)); }
#else
typedef __bf16 rn_h16;
#define RN_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define RN_DS_READ_TR16_B64 __builtin_amdgcn_ds_read_tr16_b64_v4bf16
#define RN_SIX_X2 0x40c040c0u   /* 6.0 twice */
// ---- bf16 <-> f32 (round to nearest even), device side ------------------------------------
__device__ __forceinline__ float rn_bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// v_cvt_pk_bf16_f32 (gfx950): round to nearest even, the values of the integer sequence
//   u += 0x7fff + ((u >> 16) & 1); u >>= 16
// for every finite input and the infinities; a NaN stays a quiet NaN.  The integer form cost ~7 VALU instructions per
// element and was most of the arithmetic of every elementwise bf16 kernel (the 16-byte pack of fpn_topdown_kernel's
// four chained levels: 139 us, VALU-bound).
__device__ __forceinline__ uint16_t rn_f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float rn_lo16(uint32_t u) { return __uint_as_float(u << 16); }          // low / high element
__device__ __forceinline__ float rn_hi16(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }  // of a packed pair
#endif
// round to the 16-bit type and back: where the reference materialises a 16-bit tensor between two layers.  The hardware
// conversion (v_cvt_pk_bf16_f32, round to nearest even — the same values as rn_f32_to_bf16 for every finite input)
// instead of the ~6-instruction integer sequence: the 128-row conv epilogue applies it to every accumulator.
__device__ __forceinline__ float rn_rb(float v) { return (float)(rn_h16)v; }
#ifdef RN_F16
__device__ __forceinline__ uint32_t rn_pack_bf16x2(float lo, float hi) {
  return (uint32_t)rn_f32_to_bf16(lo) | ((uint32_t)rn_f32_to_bf16(hi) << 16);
}
#else
typedef __bf16 rn_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float rn_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rn_pack_bf16x2(float lo, float hi) {   // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(rn_f32x2_t{lo, hi}, rn_bf16x2_t));
}
#endif

// Activation of N values with a wave-uniform selector.  One branch per CALL (not per element: a `switch` inside an
// unrolled element loop compiles to a chain of scalar compares and taken branches per element — measured 25 000
// cycles per 256 x 256 tile epilogue against ~4 000 for the arithmetic): swish behind one uniform branch, identity
// returns at once, relu / relu6 are max(v, 0) then min(v, 6 or +inf).
// x * sigmoid(x) with v_rcp_f32 (1 ulp) for the reciprocal: the correctly rounded division the build flags give `/` is
// ~11 instructions per element in kernels that are VALU-bound on EfficientNet (depthwise epilogue, BatchNorm apply), and the
// value is rounded to 16 bits right after.  x -> -inf gives -0 like the division did.
__device__ __forceinline__ float rn_swish(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int N>
__device__ __forceinline__ void rn_apply_act_n(float (&f)[N], int act) {
  if (act == RN_ACT_SWISH) {
#pragma unroll
    for (int q = 0; q < N; ++q) f[q] = rn_swish(f[q]);
    return;
  }
  if (act == RN_ACT_NONE) return;
  const float hi = act == RN_ACT_RELU6 ? 6.0f : __builtin_huge_valf();
#pragma unroll
  for (int q = 0; q < N; ++q) f[q] = fminf(fmaxf(f[q], 0.0f), hi);
}
__device__ __forceinline__ float rn_apply_act(float v, int act) {
  float f[1] = {v};
  rn_apply_act_n<1>(f, act);
  return f[0];
}

// 64-wide wavefront reductions
__device__ __forceinline__ float rn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double rn_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ unsigned long long rn_wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// a / b for 0 <= a < 2^22, b > 0, through the float reciprocal (rcp_b ~ 1 / b): the estimate is off by at most
// one, fixed up with the exact remainder — ~8 VALU instructions against ~45 for the compiler's integer division.
// The tile set-up code of the persistent kernels runs a few dozen of these per lane per tile, on all eight waves.
// the reciprocal estimate for rn_fdiv: v_rcp_f32 (1 ulp).  a * rcp is then within 0.75 of a / b for a < 2^22, which the
// +-1 fix-up covers; __frcp_rn (correctly rounded under -fhip-fp32-correctly-rounded-divide-sqrt) compiled to a
// 12-instruction v_div_scale / v_div_fmas / v_div_fixup sequence per call, ~60 instructions per tile set-up
__device__ __forceinline__ float rn_rcp(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ int rn_fdiv(int a, int b, float rcp_b) {
  int q = (int)((float)a * rcp_b);
  const int r = a - q * b;
  q += (r >= b ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}

// (n, y, x, 16-byte channel group) of a flat index into an [N][H][W][C8] tensor, for the elementwise NHWC kernels (one
// thread per 16 bytes).  Six integer divisions by run-time values per thread made those kernels VALU-bound: ~100
// instructions each as 64-bit divisions, ~25 as 32-bit ones; the kernels ran at ~2 TB/s.  mode (uniform, from the
// tensor's size): 0 = 64-bit divisions (2^31 elements or more), 1 = 32-bit divisions, 2 = fewer than 2^22 pixels:
// the channel group by mask / shift when C8 is a power of two, rows and images through rn_fdiv (~8 instructions each).
struct RnIdx4 { int n, y, x, c; };
__device__ __forceinline__ int rn_decode_mode(long long total, int C8) {
  return total >= (1ll << 31) ? 0 : (total < ((long long)C8 << 22) ? 2 : 1);
}
__device__ __forceinline__ RnIdx4 rn_decode4(long long t, int C8, int W, int H, int mode) {
  RnIdx4 r;
  if (mode == 2) {
    const unsigned u = (unsigned)t;
    unsigned pix;
    if ((C8 & (C8 - 1)) == 0) {
      r.c = (int)(u & (unsigned)(C8 - 1));
      pix = u >> (31 - __builtin_clz((unsigned)C8));
    } else {
      pix = u / (unsigned)C8;
      r.c = (int)(u - pix * (unsigned)C8);
    }
    const int row = rn_fdiv((int)pix, W, rn_rcp((float)W));
    r.x = (int)pix - row * W;
    r.n = rn_fdiv(row, H, rn_rcp((float)H));
    r.y = row - r.n * H;
  } else if (mode == 1) {
    unsigned u = (unsigned)t;
    r.c = (int)(u % (unsigned)C8); u /= (unsigned)C8;
    r.x = (int)(u % (unsigned)W); u /= (unsigned)W;
    r.y = (int)(u % (unsigned)H);
    r.n = (int)(u / (unsigned)H);
  } else {
    r.c = (int)(t % C8); t /= C8;
    r.x = (int)(t % W); t /= W;
    r.y = (int)(t % H);
    r.n = (int)(t / H);
  }
  return r;
}
