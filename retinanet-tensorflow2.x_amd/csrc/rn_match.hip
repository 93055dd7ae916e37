// rn_match.hip — a2-a4: pairwise IoU, anchor<->GT matching with force-match, target encode.
// Reference: retinanet/dataloader/utils.py:17-46 (compute_iou / convert_to_corners),
//            retinanet/dataloader/label_encoder.py:27-55 (_match_anchor_boxes),
//            :57-76 (_compute_box_target), :79-94 (_pad_labels + gathers), :122-124.
//
// Two HBM-bound launches per batch (the G x A IoU matrix is never materialised):
//   pass 1  (grid: anchor chunks x images) every thread owns one anchor, the image's GT boxes
//           sit in LDS.  It scans the GTs in ascending order keeping the FIRST maximum
//           (tf.argmax over axis 0, :33-34) and stores the provisional match
//           {idx | -1 | -2} (:36-42).  For the per-GT argmax over anchors (:44-46) each wave
//           reduces a 64-bit key (iou_bits << 32 | ~anchor) with max -> highest IoU, lowest
//           anchor index on ties; a workgroup (1024 anchors, 4 per thread) combines its waves' keys in LDS and
//           folds ONE atomicMax per (workgroup, GT) into best[b][g] (one per (wave, GT) queued ~1200 atomics on
//           each address: 414 us for 32 images).
//           IoU is in [0,1] so its bit pattern orders like the value; atomicMax is order
//           independent, so the result is deterministic.
//   pass 2  (same grid) loads best[b][*] into LDS; an anchor that is some GT's best anchor
//           takes the LOWEST such GT index (argmax over the one-hot, :47-54); then gathers
//           the GT (or the two sentinel rows of _pad_labels) and writes class/box targets;
//           positives are counted with integer atomics (deterministic), one per workgroup, and converted
//           to f32 by match_finalize (the first version's per-workgroup fence + ticket cost 330 us).
// Compiled with -ffp-contract=off: every product below is rounded before it is added, as in
// the reference's separate TF ops.  Algorithmic bytes per (image, anchor): 16 B anchor read
// in each pass + 4 B match write/read + 4+4+16 B outputs = 64 B.
#include "rn_common.h"
#include "../../include/rn_math.h"

#define RN_MATCH_THREADS 256
#define RN_MATCH_GMAX 1024

__device__ __forceinline__ float iou_cxcywh(float4 g, float4 a) {
  // convert_to_corners (utils.py:17-24): xy -/+ wh / 2.0
  const float gx1 = g.x - g.z / 2.0f, gy1 = g.y - g.w / 2.0f;
  const float gx2 = g.x + g.z / 2.0f, gy2 = g.y + g.w / 2.0f;
  const float ax1 = a.x - a.z / 2.0f, ay1 = a.y - a.w / 2.0f;
  const float ax2 = a.x + a.z / 2.0f, ay2 = a.y + a.w / 2.0f;
  const float lux = fmaxf(gx1, ax1), luy = fmaxf(gy1, ay1);
  const float rdx = fminf(gx2, ax2), rdy = fminf(gy2, ay2);
  const float iw = fmaxf(0.0f, rdx - lux), ih = fmaxf(0.0f, rdy - luy);
  const float inter = iw * ih;
  const float area_g = g.z * g.w;
  const float area_a = a.z * a.w;
  const float uni = fmaxf(area_g + area_a - inter, 1e-8f);
  return fminf(fmaxf(inter / uni, 0.0f), 1.0f);
}

// anchors per thread: a workgroup covers RN_MATCH_APT * 256 consecutive anchors of one image (anchor
// k * 256 + thread of the block's range), so the per-(image, GT) atomics below are issued once per 1024 anchors
#define RN_MATCH_APT 4

__global__ void __launch_bounds__(RN_MATCH_THREADS)
match_pass1(const float4* __restrict__ anchors, long long A, const float4* __restrict__ gt_boxes,
            const int* __restrict__ gt_counts, int Gmax, float match_iou, float ignore_iou,
            int* __restrict__ matches, unsigned long long* __restrict__ best) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* sgt = (float4*)smem;
  unsigned long long* sbest = (unsigned long long*)(smem + (size_t)(Gmax < 1 ? 1 : Gmax) * sizeof(float4));
  const int b = blockIdx.y;
  int G = gt_counts[b];
  G = G < 0 ? 0 : (G > Gmax ? Gmax : G);
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    sgt[g] = gt_boxes[(long long)b * Gmax + g];
    sbest[g] = 0ull;
  }
  __syncthreads();
  const long long a0 = (long long)blockIdx.x * (RN_MATCH_APT * RN_MATCH_THREADS) + threadIdx.x;
  float4 an[RN_MATCH_APT];
  bool live[RN_MATCH_APT];
  float max_iou[RN_MATCH_APT];   // any real IoU (>= 0) beats -1; stays -1 only when G == 0
  int arg[RN_MATCH_APT];
#pragma unroll
  for (int k = 0; k < RN_MATCH_APT; ++k) {
    const long long a_idx = a0 + k * RN_MATCH_THREADS;
    live[k] = a_idx < A;
    an[k] = live[k] ? anchors[a_idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    max_iou[k] = -1.0f;
    arg[k] = 0;
  }
  for (int g = 0; g < G; ++g) {
    const float4 gt = sgt[g];
    unsigned long long key = 0ull;
#pragma unroll
    for (int k = 0; k < RN_MATCH_APT; ++k) {
      const float v = live[k] ? iou_cxcywh(gt, an[k]) : 0.0f;
      if (v > max_iou[k]) {
        max_iou[k] = v;
        arg[k] = g;
      }
      const unsigned int inv = ~(unsigned int)(a0 + k * RN_MATCH_THREADS);
      const unsigned long long kk = live[k] ? (((unsigned long long)__float_as_uint(v)) << 32) | inv : 0ull;
      key = kk > key ? kk : key;
    }
    // workgroup maximum through LDS (the wave maximum first: one LDS atomic per wave), one global atomic per
    // (workgroup, GT) after the scan — with one per (wave, GT) ~1200 waves queued on each best[b][g]
    key = rn_wave_max_u64(key);
    if ((threadIdx.x & 63) == 0 && key != 0ull) atomicMax(&sbest[g], key);
  }
#pragma unroll
  for (int k = 0; k < RN_MATCH_APT; ++k) {
    if (!live[k]) continue;
    int m = -1;
    if (G > 0) {
      m = (max_iou[k] > match_iou) ? arg[k] : -1;
      if (max_iou[k] >= ignore_iou && match_iou > max_iou[k]) m = -2;
    }
    matches[(long long)b * A + a0 + k * RN_MATCH_THREADS] = m;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += blockDim.x)
    if (sbest[g] != 0ull) atomicMax(&best[(long long)b * Gmax + g], sbest[g]);
}

__global__ void __launch_bounds__(RN_MATCH_THREADS)
match_pass2(const float4* __restrict__ anchors, long long A, const float4* __restrict__ gt_boxes,
            const float* __restrict__ gt_classes, const int* __restrict__ gt_counts, int Gmax,
            const unsigned long long* __restrict__ best, int* __restrict__ matches,
            float* __restrict__ class_targets, float4* __restrict__ box_targets, int* __restrict__ pos_count,
            float4 inv_var, int use_var) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned int* sbest = (unsigned int*)smem;
  int& s_pos = *(int*)(smem + (size_t)(Gmax < 1 ? 1 : Gmax) * 4);
  const int b = blockIdx.y;
  int G = gt_counts[b];
  G = G < 0 ? 0 : (G > Gmax ? Gmax : G);
  if (threadIdx.x == 0) s_pos = 0;
  for (int g = threadIdx.x; g < G; g += blockDim.x)
    sbest[g] = ~(unsigned int)(best[(long long)b * Gmax + g] & 0xffffffffull);
  __syncthreads();
  int positives = 0;
#pragma unroll
  for (int k = 0; k < RN_MATCH_APT; ++k) {
    const long long a_idx = (long long)blockIdx.x * (RN_MATCH_APT * RN_MATCH_THREADS) + k * RN_MATCH_THREADS + threadIdx.x;
    if (a_idx >= A) continue;
    int m = matches[(long long)b * A + a_idx];
    for (int g = 0; g < G; ++g) {
      if (sbest[g] == (unsigned int)a_idx) {
        m = g;
        break;
      }
    }
    const float4 an = anchors[a_idx];
    float4 gt = make_float4(0.f, 0.f, 0.f, 0.f);  // rows 0/1 of _pad_labels are zero boxes
    float cls = (m == -2) ? -2.0f : -1.0f;
    if (m >= 0) {
      gt = gt_boxes[(long long)b * Gmax + m];
      cls = gt_classes[(long long)b * Gmax + m];
    }
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m >= 0) {
      const float eps = 1e-8f;  // label_encoder.py:57-58, applied to all four coordinates
      const float gx = fmaxf(gt.x, eps), gy = fmaxf(gt.y, eps), gw = fmaxf(gt.z, eps), gh = fmaxf(gt.w, eps);
      t.x = (gx - an.x) / an.z;
      t.y = (gy - an.y) / an.w;
      t.z = rn_logf(gw / an.z);
      t.w = rn_logf(gh / an.w);
      if (use_var) {
        t.x = t.x / inv_var.x;
        t.y = t.y / inv_var.y;
        t.z = t.z / inv_var.z;
        t.w = t.w / inv_var.w;
      }
      ++positives;
    }
    matches[(long long)b * A + a_idx] = m;
    class_targets[(long long)b * A + a_idx] = cls;
    box_targets[(long long)b * A + a_idx] = t;
  }
  // positives: integer sums (deterministic); the float conversion is match_finalize's
  if (positives) atomicAdd(&s_pos, positives);
  __syncthreads();
  if (threadIdx.x == 0 && s_pos) atomicAdd(&pos_count[b], s_pos);
}

__global__ void match_finalize(const int* __restrict__ pos_count, float* __restrict__ num_positives, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) num_positives[b] = (float)pos_count[b];
}

extern "C" size_t rn_match_workspace_bytes(int B, int Gmax) {
  if (B <= 0) return 0;
  if (Gmax < 1) Gmax = 1;
  return rn_align_up((size_t)B * Gmax * 8, 256) + rn_align_up((size_t)B * 4, 256) * 2;
}

extern "C" int rn_anchor_match_encode(const float* anchors, int64_t A, const float* gt_boxes,
                                      const float* gt_classes, const int32_t* gt_counts, int B, int Gmax,
                                      float match_iou, float ignore_iou, const float* box_variance,
                                      int32_t* matches, float* class_targets, float* box_targets,
                                      float* num_positives, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  RN_CHECK_ARG(anchors && gt_counts && matches && class_targets && box_targets && num_positives,
               "rn_anchor_match_encode: null argument");
  RN_CHECK_ARG(A > 0 && A < (1ll << 31) && B > 0, "rn_anchor_match_encode: bad A=%lld B=%d", (long long)A, B);
  RN_CHECK_ARG(Gmax >= 0 && Gmax <= RN_MATCH_GMAX, "rn_anchor_match_encode: Gmax=%d outside 0..%d", Gmax,
               RN_MATCH_GMAX);
  RN_CHECK_ARG(Gmax == 0 || (gt_boxes && gt_classes), "rn_anchor_match_encode: null gt arrays");
  const size_t need = rn_match_workspace_bytes(B, Gmax);
  if (workspace_bytes < need || !workspace) {
    rn_set_error("rn_anchor_match_encode: workspace %zu < %zu", workspace_bytes, need);
    return RN_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream;
  const int Gm = Gmax < 1 ? 1 : Gmax;
  char* ws = (char*)workspace;
  unsigned long long* best = (unsigned long long*)ws;
  int* pos = (int*)(ws + rn_align_up((size_t)B * Gm * 8, 256));
  RN_CHECK_HIP(hipMemsetAsync(workspace, 0, need, st));
  dim3 grid((unsigned)rn_cdiv(A, RN_MATCH_THREADS * RN_MATCH_APT), (unsigned)B);
  const size_t lds1 = (size_t)Gm * (sizeof(float4) + sizeof(unsigned long long));
  hipLaunchKernelGGL(match_pass1, grid, dim3(RN_MATCH_THREADS), lds1, st, (const float4*)anchors, (long long)A,
                     (const float4*)gt_boxes, gt_counts, Gmax, match_iou, ignore_iou, matches, best);
  RN_CHECK_LAUNCH();
  float4 var = make_float4(1.f, 1.f, 1.f, 1.f);
  if (box_variance) var = make_float4(box_variance[0], box_variance[1], box_variance[2], box_variance[3]);
  hipLaunchKernelGGL(match_pass2, grid, dim3(RN_MATCH_THREADS), (size_t)Gm * 4 + 16, st, (const float4*)anchors,
                     (long long)A, (const float4*)gt_boxes, gt_classes, gt_counts, Gmax, best, matches,
                     class_targets, (float4*)box_targets, pos, var, box_variance ? 1 : 0);
  RN_CHECK_LAUNCH();
  hipLaunchKernelGGL(match_finalize, dim3((unsigned)rn_cdiv(B, 256)), dim3(256), 0, st, pos, num_positives, B);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
