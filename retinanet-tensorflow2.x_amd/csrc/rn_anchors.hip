// rn_anchors.hip — a1: anchor box generation (retinanet/dataloader/anchor_generator.py:24-104).
//
// One thread per anchor row.  Row order: level -> y -> x -> anchor, anchor = ratio*num_scales
// + scale (anchor_generator.py:55-62), each row [cx, cy, w, h] (anchor_generator.py:86-88).
// Arithmetic follows the reference op by op in fp32 with correctly rounded sqrt and divide:
//   h = sqrt(f32(area/ratio)); w = f32(area)/h; dims = f32(scale) * [w, h]   (:55-61)
//   cx = (x + 0.5) * stride;  cy = (y + 0.5) * stride                        (:78-81)
// HBM-bound, write-only: 16 B per anchor row.
#include "rn_common.h"

#define RN_MAX_LEVELS 8

struct AnchorParams {
  int num_levels, num_ratios, num_scales;
  int fh[RN_MAX_LEVELS], fw[RN_MAX_LEVELS];
  float stride[RN_MAX_LEVELS];
  long long begin[RN_MAX_LEVELS + 1];
  float areas[RN_MAX_LEVELS];
  float aor[RN_MAX_LEVELS * 8];
  float scales[8];
};

__global__ void __launch_bounds__(256) anchors_kernel(float4* __restrict__ boxes, AnchorParams p) {
  const long long total = p.begin[p.num_levels];
  const int A = p.num_ratios * p.num_scales;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < p.num_levels && i >= p.begin[l + 1]) ++l;
    const long long j = i - p.begin[l];
    const int a = (int)(j % A);
    const long long cell = j / A;
    const int x = (int)(cell % p.fw[l]);
    const int y = (int)(cell / p.fw[l]);
    const int ri = a / p.num_scales, si = a % p.num_scales;
    const float h = __fsqrt_rn(p.aor[l * p.num_ratios + ri]);
    const float w = __fdiv_rn(p.areas[l], h);
    const float s = p.scales[si];
    float4 o;
    o.x = ((float)x + 0.5f) * p.stride[l];
    o.y = ((float)y + 0.5f) * p.stride[l];
    o.z = s * w;
    o.w = s * h;
    boxes[i] = o;
  }
}

extern "C" int rn_anchors_generate(float* boxes, int64_t cap_rows, int img_h, int img_w, int min_level,
                                   int max_level, const float* areas, const float* area_over_ratio,
                                   int num_ratios, const float* scales, int num_scales, int64_t* n_out,
                                   void* stream) {
  const int L = max_level - min_level + 1;
  RN_CHECK_ARG(L >= 1 && L <= RN_MAX_LEVELS, "rn_anchors_generate: levels %d..%d unsupported", min_level,
               max_level);
  RN_CHECK_ARG(num_ratios >= 1 && num_ratios <= 8 && num_scales >= 1 && num_scales <= 8,
               "rn_anchors_generate: num_ratios/num_scales must be in 1..8");
  RN_CHECK_ARG(areas && area_over_ratio && scales && n_out, "rn_anchors_generate: null argument");
  AnchorParams p;
  p.num_levels = L;
  p.num_ratios = num_ratios;
  p.num_scales = num_scales;
  p.begin[0] = 0;
  for (int l = 0; l < L; ++l) {
    const int lv = min_level + l;
    const int d = 1 << lv;
    p.fh[l] = (img_h + d - 1) / d;  // ceil(H / 2^l), anchor_generator.py:45-47,99-100
    p.fw[l] = (img_w + d - 1) / d;
    p.stride[l] = (float)d;
    p.begin[l + 1] = p.begin[l] + (long long)p.fh[l] * p.fw[l] * num_ratios * num_scales;
    p.areas[l] = areas[l];
    for (int r = 0; r < num_ratios; ++r) p.aor[l * num_ratios + r] = area_over_ratio[l * num_ratios + r];
  }
  for (int s = 0; s < num_scales; ++s) p.scales[s] = scales[s];
  *n_out = p.begin[L];
  if (boxes == nullptr) return RN_OK;  // size query
  RN_CHECK_ARG(cap_rows >= p.begin[L], "rn_anchors_generate: boxes has %lld rows, need %lld",
               (long long)cap_rows, p.begin[L]);
  const int blocks = (int)(rn_cdiv(p.begin[L], 256) < 2048 ? rn_cdiv(p.begin[L], 256) : 2048);
  hipLaunchKernelGGL(anchors_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)boxes, p);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
