// rn_conv_dev.h — device-side declarations shared by the implicit-GEMM conv kernels
// (rn_conv.hip: 128-row tiles; rn_conv_big.hip: 256 x 256 tiles).
#ifndef RN_CONV_DEV_H_
#define RN_CONV_DEV_H_
#include "rn_common.h"

typedef rn_h16 bf16x8_t __attribute__((ext_vector_type(8)));   // 8 elements of the 16-bit storage type (rn_common.h)
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CONV_THREADS 256

struct ConvSegDev {
  const uint16_t* x;
  const uint16_t* w;
  void* y;
  const float* scale;
  const float* shift;
  const uint16_t* residual;
  float* bn_partial;   // fused BatchNorm forward statistics (bf16 outputs), or null
  const float* bias;   // Conv2D bias, added to the fp32 accumulator (rnet_hip.h: epilogue)
  const uint16_t* bn_y;   // fused stage 1 of the BatchNorm backward reduction (rn_conv_segment.bn_bwd_y), or null
  const float* bn_fwd;    //   that layer's mean | invstd | scale | shift
  int N, H, W, Cin, pix_stride, Ho, Wo, Cout;
  int M, tile_begin, n_tiles, CinP;  // CinP = K extent per tap = w_terms * (Cin rounded up to the K step)
  int cwrap, halo_pitch;             // input channels wrap at cwrap (= CinP / w_terms): split-bf16 weight planes;
                                     // halo kernel: patch pixels per image row (W + 1, or that rounded up to 8)
  int pair_cout, rows;               // rn_conv_segment.w_pair: Cout above = the GEMM's columns (rn_conv_pair_rows), this =
                                     // the channels of y / bias; 0 = off (256- / 512-row kernels, f32 epilogue only)
                                     // rows (conv_big_kernel): output pixels a tile really covers, 0 = all 256 — balanced
                                     // tiles of the HBM-bound 1x1 layers (rn_conv.hip: conv_big_balanced_rows)
};


struct ConvArgs {
  int R, S, sh, sw, pt, pl, act, nseg, total_tiles, pad_;
  // split-K of the last round of the persistent 256-row kernels (rnet_hip.h: rn_conv_problem.splitk_ws): the tiles
  // [split_f, total_tiles) run in a second, SPLIT launch of vtotal = (total_tiles - split_f) * split_s workgroups;
  // workgroup l * split_s + part computes the channel chunks [part * nch / split_s, (part + 1) * nch / split_s) of tile
  // split_f + l.  No split: split_s = 1.
  int split_f, split_s, vtotal, pad2_;
  float* ws;   // [4096 words: arrival counter (leftover tile l, wave w) at word l * 8 + w][slot (l, part): 65536 floats]
  ConvSegDev seg[RN_CONV_MAX_SEGMENTS];
};

#define RN_SPLITK_HEADER_BYTES 16384
#define RN_SPLITK_SLOT_BYTES (256 * 256 * 4)

template <int BK>
__device__ __forceinline__ int lds_swz(int row) {
  constexpr int SLOTS = BK / 8;        // 16-byte slots per row
  constexpr int RPB = 256 / (BK * 2);  // rows per 256-byte bank row
  return (row / RPB) % SLOTS;
}
template <int BK>
__device__ __forceinline__ int lds_slot_off(int row, int slot) {
  return (row * (BK / 8) + (slot ^ lds_swz<BK>(row))) * 16;
}

typedef __attribute__((address_space(3))) void lds_void_t;

// 16-byte buffer load straight into LDS (no VGPR round trip).  LDS address = M0 base (wave
// uniform) + lane*16; the global source offset is per lane; out-of-range offsets write zeros.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}

// (rn_rcp / rn_fdiv: rn_common.h)
#define RN_OOB 0x80000000u


// rn_conv_big.hip
int rn_launch_conv_big(const ConvArgs& a, bool out_f32, const rn_launch_opts& opts, hipStream_t st);   // (bn_y set on segment 0: the BN_BWD variant)
// rn_conv_halo.hip (3x3 / stride 1 / pad 1)
int rn_launch_conv_halo(const ConvArgs& a, bool out_f32, const rn_launch_opts& opts, hipStream_t st, int wm = 2);   // wm = 4: 512 x 128 tiles
// plan of the last-round split for a persistent launch of `total_tiles` tiles whose shortest tile has `min_chunks`
// K chunks; fills a.split_f / split_s / vtotal / ws (no split when ws is null or too small) and returns the grid
int rn_splitk_plan(ConvArgs& a, int min_chunks, void* ws, long long ws_bytes, const rn_launch_opts& opts);
int rn_conv_halo_patch_pixels(int N, int H, int W, int pitch, int BM = 256);
int rn_conv_halo_pitch(int W);
int rn_conv_halo_capacity(int BM = 256);
#endif  // RN_CONV_DEV_H_
