/* rn_oracle.c — CPU restatement (plain C) of the sequential parts of the reference's
 * detection post-processing and of the shared fp32 transcendental functions.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (retinanet-tensorflow2.x_amd/) may
 * load this file's library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do, and only as the checker.
 *
 * PARITY UNPINNED: the reference (srihari-humbarwadi/retinanet-tensorflow2.x) has no tests
 * or golden vectors and cannot be imported here (every module imports TensorFlow, which is
 * absent).  The functions below restate
 *   - tf.raw_ops.NonMaxSuppressionV5 as called from
 *     retinanet/model/layers/postprocessing_ops.py:443-451 (per class) — algorithm restated
 *     from TensorFlow 2.8's published kernel semantics (priority queue ordered by score then
 *     lower index, suppress_begin_index, Gaussian soft-NMS weight, padding with zeros);
 *   - the exp/log/sigmoid used by postprocessing_ops.py:99,114, label_encoder.py:64 and
 *     loss_impl.py:19,23, through the deterministic include/rn_math.h forms.
 * Build: oracle/build_oracle.py (gcc -O2 -ffp-contract=off).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rn_math.h"

void rn_o_expf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = rn_expf(x[i]); }
void rn_o_logf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = rn_logf(x[i]); }
void rn_o_sigmoidf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = rn_sigmoidf(x[i]); }
void rn_o_log1pf_pos(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = rn_log1pf_pos(x[i]); }

/* TensorFlow non_max_suppression IOU: corners in any order, degenerate boxes -> 0. */
static float nms_iou(const float* boxes, int i, int j) {
  const float* a = boxes + 4 * (int64_t)i;
  const float* b = boxes + 4 * (int64_t)j;
  const float ay0 = fminf(a[0], a[2]), ax0 = fminf(a[1], a[3]);
  const float ay1 = fmaxf(a[0], a[2]), ax1 = fmaxf(a[1], a[3]);
  const float by0 = fminf(b[0], b[2]), bx0 = fminf(b[1], b[3]);
  const float by1 = fmaxf(b[0], b[2]), bx1 = fmaxf(b[1], b[3]);
  const float area_a = (ay1 - ay0) * (ax1 - ax0);
  const float area_b = (by1 - by0) * (bx1 - bx0);
  if (area_a <= 0.0f || area_b <= 0.0f) return 0.0f;
  const float iy0 = fmaxf(ay0, by0), ix0 = fmaxf(ax0, bx0);
  const float iy1 = fminf(ay1, by1), ix1 = fminf(ax1, bx1);
  const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
  return inter / (area_a + area_b - inter);
}

typedef struct { int box_index; float score; int suppress_begin_index; } cand_t;

/* "less" of the max-heap: lower score, or equal score and HIGHER index, has lower priority */
static int cand_less(const cand_t* a, const cand_t* b) {
  return ((a->score == b->score) && (a->box_index > b->box_index)) || a->score < b->score;
}
static void heap_push(cand_t* h, int* n, cand_t c) {
  int i = (*n)++;
  h[i] = c;
  while (i > 0) {
    int p = (i - 1) / 2;
    if (cand_less(&h[p], &h[i])) { cand_t t = h[p]; h[p] = h[i]; h[i] = t; i = p; } else break;
  }
}
static cand_t heap_pop(cand_t* h, int* n) {
  cand_t top = h[0];
  h[0] = h[--(*n)];
  int i = 0;
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < *n && cand_less(&h[m], &h[l])) m = l;
    if (r < *n && cand_less(&h[m], &h[r])) m = r;
    if (m == i) break;
    cand_t t = h[m]; h[m] = h[i]; h[i] = t; i = m;
  }
  return top;
}

/* NonMaxSuppressionV5(boxes[n,4], scores[n], max_output_size, iou_threshold,
 * score_threshold, soft_nms_sigma, pad_to_max_output_size=True)
 * -> selected_indices[max_out] (0 padded), selected_scores[max_out] (0 padded), returns
 * the number of valid outputs. */
int rn_o_nms_v5(const float* boxes, const float* scores, int n, int max_out, float iou_threshold,
                float score_threshold, float soft_nms_sigma, int32_t* sel_idx, float* sel_scores) {
  cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
  int hn = 0;
  for (int i = 0; i < n; ++i)
    if (scores[i] > score_threshold) { cand_t c = {i, scores[i], 0}; heap_push(heap, &hn, c); }
  const int is_soft = soft_nms_sigma > 0.0f;
  const float scale = is_soft ? -0.5f / soft_nms_sigma : 0.0f;
  int nsel = 0;
  while (nsel < max_out && hn > 0) {
    cand_t next = heap_pop(heap, &hn);
    const float original = next.score;
    int hard = 0;
    for (int j = nsel - 1; j >= next.suppress_begin_index; --j) {
      const float sim = nms_iou(boxes, next.box_index, sel_idx[j]);
      /* suppress_weight: exp(scale*sim*sim) if soft or sim <= thr, else 0 */
      const float w = (is_soft || sim <= iou_threshold) ? rn_expf(scale * sim * sim) : 0.0f;
      next.score = next.score * w;
      if (!is_soft && sim > iou_threshold) { hard = 1; break; }
      if (next.score <= score_threshold) break;
    }
    next.suppress_begin_index = nsel;
    if (!hard) {
      if (next.score == original) {
        sel_idx[nsel] = next.box_index;
        sel_scores[nsel] = next.score;
        ++nsel;
        continue;
      }
      if (next.score > score_threshold) heap_push(heap, &hn, next);
    }
  }
  for (int i = nsel; i < max_out; ++i) { sel_idx[i] = 0; sel_scores[i] = 0.0f; }
  free(heap);
  return nsel;
}
