/* rnet_hip.h — C ABI of librnet_hip.so, the MI355X (gfx950) hot path that sits behind the
 * `retinanet` Python surface of srihari-humbarwadi/retinanet-tensorflow2.x.
 *
 * The reference has no FFI of its own: every entry point below replaces
 * the TensorFlow op kernels one reference function dispatches to, and cites that function
 * (paths relative to the reference checkout).  Conventions:
 *   - all functions return 0 on success or a negative rn_status; rn_last_error() returns a
 *     thread-local message for the last failure; no C++ exception crosses the boundary;
 *   - the caller owns every buffer; pointers are DEVICE pointers unless a parameter says
 *     "host"; tensors are dense, NHWC where spatial, 16-byte aligned;
 *   - every launch takes an explicit hipStream_t (passed as void*) and is asynchronous;
 *   - nothing here allocates device memory: workspaces are passed in, and each
 *     `*_workspace_bytes` function says how much a call needs.
 */
#ifndef RNET_HIP_H_
#define RNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the functions declared here are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef enum {
  RN_OK = 0,
  RN_EINVAL = -1, /* bad shape / dtype / alignment / null pointer */
  RN_ENOMEM = -2, /* workspace too small */
  RN_EHIP = -3,   /* a HIP runtime call or launch failed (message has hipGetErrorString) */
  RN_ECOMM = -4,  /* collective failure */
  RN_EUNSUPPORTED = -5 /* well-formed input of a kind this library does not implement (rn_jpeg_*: progressive / arithmetic
                        * / 12-bit / CMYK / unusual sampling): callers may route it elsewhere; RN_EINVAL = corrupt input */
} rn_status;

enum { RN_DT_F32 = 0, RN_DT_BF16 = 1 };
enum { RN_ACT_NONE = 0, RN_ACT_RELU = 1, RN_ACT_RELU6 = 2, RN_ACT_SWISH = 3 };

const char* rn_last_error(void);
/* library ABI version; bumped when a signature changes */
int rn_abi_version(void);
/* 16-bit storage type of activations / packed weights this build was compiled for: 0 = bfloat16 (librnet_hip.so,
 * `mixed_bfloat16`), 1 = IEEE half (librnet_hip_f16.so: the same sources with -DRN_F16, `mixed_float16`).  Wherever this
 * header says "bf16" for a tensor, the half build stores IEEE half there; RN_DT_BF16 names "the 16-bit type". */
int rn_storage_dtype(void);
/* 1 if a gfx950 device is visible to this process, 0 otherwise (never initialises more
 * than hipGetDeviceCount + properties) */
int rn_device_ok(void);

/* ---------------------------------------------------------------------------------------
 * a1  AnchorBoxGenerator  (retinanet/dataloader/anchor_generator.py:24-104)
 * out: boxes f32[n_total,4] rows [cx,cy,w,h], ordered level -> y -> x -> (ratio-major,
 * scale-minor) anchor.  area_over_ratio is host f32[num_levels*num_ratios] =
 * f32(area/ratio) (the reference does that division in Python floats, :56), areas host
 * f32[num_levels], scales host f32[num_scales].  Returns n_total through *n_out (host).
 */
int rn_anchors_generate(float* boxes, int64_t boxes_capacity_rows, int img_h, int img_w, int min_level,
                        int max_level, const float* areas, const float* area_over_ratio, int num_ratios,
                        const float* scales, int num_scales, int64_t* n_out, void* stream);

/* ---------------------------------------------------------------------------------------
 * a2-a4  compute_iou + LabelEncoder._match_anchor_boxes/_compute_box_target/encode_sample
 * (retinanet/dataloader/utils.py:27-46, retinanet/dataloader/label_encoder.py:27-125)
 * in : anchors f32[A,4] cxcywh; gt_boxes f32[B,Gmax,4] cxcywh pixels; gt_classes
 *      f32[B,Gmax]; gt_counts i32[B] (0 <= count <= Gmax)
 * out: matches i32[B,A] in {-2,-1,0..G-1}; class_targets f32[B,A] (class id, -1
 *      background, -2 ignore); box_targets f32[B,A,4]; num_positives f32[B]
 * box_variance: host f32[4] or NULL (encoder_params.scale_box_targets, :73-76).
 * workspace: rn_match_workspace_bytes(B, Gmax) bytes.
 */
size_t rn_match_workspace_bytes(int B, int Gmax);
int rn_anchor_match_encode(const float* anchors, int64_t A, const float* gt_boxes, const float* gt_classes,
                           const int32_t* gt_counts, int B, int Gmax, float match_iou, float ignore_iou,
                           const float* box_variance, int32_t* matches, float* class_targets,
                           float* box_targets, float* num_positives, void* workspace, size_t workspace_bytes,
                           void* stream);

/* ---------------------------------------------------------------------------------------
 * a9  RetinaNetLoss / ClassLoss / FocalLossV1 / BoxLoss
 * (retinanet/losses/retinanet_loss.py:37-83, retinanet/losses/loss_impl.py:15-28,41-105)
 * Fused forward + backward over all pyramid levels in one launch.
 * level l (0..num_levels-1): class_logits[l] f32[B, n_l, K], box_preds[l] f32[B, n_l, 4]
 * (n_l = anchors on that level; pointers are host arrays of device pointers),
 * level_offsets host i64[num_levels+1] = anchor boundaries.  Targets are the flattened
 * outputs of rn_anchor_match_encode.  normalizer: device f32[1] (already all-reduced and
 * divided by replicas, retinanet_loss.py:46-49).  grad_scale multiplies both gradients
 * (1/num_replicas, times the loss scale under fp16: executor.py:421-425).
 * out: losses f32[4] = {box-loss, class-loss, weighted-loss, normalizer}; d_class_logits[l],
 * d_box_preds[l] same shapes as the inputs (may be NULL arrays to skip the backward).
 */
size_t rn_loss_workspace_bytes(int B, int64_t A, int K);
int rn_retinanet_loss_fwd_bwd(const float* const* class_logits, const float* const* box_preds,
                              float* const* d_class_logits, float* const* d_box_preds,
                              const int64_t* level_offsets, int num_levels, int B, int K,
                              const float* class_targets, const float* box_targets, const float* normalizer,
                              float alpha, float gamma, float label_smoothing, float delta,
                              float box_loss_weight, float class_loss_weight, float grad_scale, float* losses,
                              void* workspace, size_t workspace_bytes, void* stream);
/* Same losses; the gradients are written as bf16 (round to nearest even of the same fp32 values) straight into the
 * NHWC tensors the prediction convs' backward pass reads: d_class_bf16[l] bf16[B*H_l*W_l][class_pix_stride] with the
 * anchors_per_location*K live channels in front (the pad channels are not touched), d_box_bf16[l] likewise with
 * anchors_per_location*4.  Saves the fp32 gradient round trip of executor.py:427 (tape.gradient keeps fp32). */
int rn_retinanet_loss_fwd_bwd_bf16(const float* const* class_logits, const float* const* box_preds,
                                   void* const* d_class_bf16, void* const* d_box_bf16, int class_pix_stride,
                                   int box_pix_stride, int anchors_per_location, const int64_t* level_offsets,
                                   int num_levels, int B, int K, const float* class_targets,
                                   const float* box_targets, const float* normalizer, float alpha, float gamma,
                                   float label_smoothing, float delta, float box_loss_weight,
                                   float class_loss_weight, float grad_scale, float* losses, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a11+a12  FuseDetections + TransformBoxesAndScores
 * (retinanet/model/layers/postprocessing_ops.py:15-56, 87-117)
 * Box decode only: the sigmoid of a12 is fused into rn_detect_per_class / rn_sigmoid_scores.
 * in : box_preds[l] f32[B,n_l,4] per level, anchors f32[A,4]
 * out: boxes f32[B,A,4] = [x1,y1,x2,y2] / [W? see note] normalised by (input_h,input_w,input_h,
 *      input_w) exactly as the reference divides (postprocessing_ops.py:65-69,104).
 */
int rn_decode_boxes(const float* const* box_preds, const int64_t* level_offsets, int num_levels, int B,
                    const float* anchors, const float* box_variance, float input_h, float input_w,
                    float* boxes, void* stream);

/* scores = sigmoid(logits) flattened over levels: f32[B,A,K] (a12, :114). */
int rn_sigmoid_scores(const float* const* class_logits, const int64_t* level_offsets, int num_levels, int B,
                      int K, float* scores, void* stream);

/* ---------------------------------------------------------------------------------------
 * a13  FilterTopKDetections._filter_per_class  (postprocessing_ops.py:128-147)
 * in : scores f32[B,A,K]; out: topk_scores f32[B,k,K], topk_indices i32[B,k,K] (anchor ids),
 * canonical order: descending score, ties by ascending anchor index.
 * k = min(top_k, A).
 */
size_t rn_topk_workspace_bytes(int B, int64_t A, int K);
int rn_topk_per_class(const float* scores, int B, int64_t A, int K, int top_k, float* topk_scores,
                      int32_t* topk_indices, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * a12+a13+a14 fused detection post-process: TransformBoxesAndScores -> FilterTopKDetections
 * -> GenerateDetections._per_class_nms  (postprocessing_ops.py:107-117,128-147,434-535)
 * in : class_logits[l] f32[B,n_l,K] per level; boxes f32[B,A,4] from rn_decode_boxes
 * out: det_boxes f32[B,max_det,4], det_scores f32[B,max_det], det_classes i32[B,max_det],
 *      valid i32[B].  soft_nms_sigma == 0 -> PerClassHardNMS, > 0 -> PerClassSoftNMS (the
 *      reference passes sigma/2 to NonMaxSuppressionV5 and iou_threshold 1.0, :448-450;
 *      pass the CONFIG sigma here).  pre_nms_top_k <= 0 disables the top-k filter.
 *      Soft NMS keeps a whole candidate list on chip: it needs 0 < pre_nms_top_k <= 8192 (or at most 8192
 *      anchors); lists of up to 5120 candidates (every shipped configuration: 5000) run the one-step-per-
 *      selection kernel, longer ones the queue loop — the same selections and score bits either way
 *      (RN_EINVAL beyond 8192).  max_det <= 256, K * max_det <= 16384.
 */
size_t rn_detect_workspace_bytes(int B, int64_t A, int K, int max_det);
int rn_detect_per_class(const float* const* class_logits, const int64_t* level_offsets, int num_levels, int B,
                        int K, const float* boxes, int pre_nms_top_k, float iou_threshold,
                        float score_threshold, float soft_nms_sigma, int max_det, float* det_boxes,
                        float* det_scores, int32_t* det_classes, int32_t* valid, void* workspace,
                        size_t workspace_bytes, void* stream);

/* a14 stand-alone: per-class NMS on already filtered candidates
 * in : cand_scores f32[B,n,K], cand_boxes f32[B,n,K,4] (class-specific boxes, as
 * FilterTopKDetections emits them); same outputs as rn_detect_per_class. */
size_t rn_nms_workspace_bytes(int B, int n, int K, int max_det);
int rn_nms_per_class(const float* cand_scores, const float* cand_boxes, int B, int n, int K,
                     float iou_threshold, float score_threshold, float soft_nms_sigma, int max_det,
                     float* det_boxes, float* det_scores, int32_t* det_classes,
                     int32_t* det_index /* i32[B,max_det] candidate row of each detection, or NULL */,
                     int32_t* valid, void* workspace, size_t workspace_bytes, void* stream);

/* a15  GenerateDetections._global_nms helper (postprocessing_ops.py:248-261): per row of
 * scores f32[rows,K] the maximum and the FIRST argmax (tf.reduce_max / tf.argmax). */
int rn_rowmax_argmax(const float* scores, int64_t rows, int K, float* max_out, int32_t* argmax_out, void* stream);

/* ---------------------------------------------------------------------------------------
 * per-device handle and per-call launch options.
 *
 * The library keeps NO mutable process-wide state: what used to be process-global tuning / test knobs is an
 * rn_launch_opts value carried by every MFMA problem descriptor (rn_conv_problem.opts, rn_wgrad_problem.opts), so two
 * engines in one process (train + eval of one Executor) cannot see each other's settings and a launch is a pure
 * function of its arguments.  All zero = the dispatcher's own choice.
 */
typedef struct {
  int32_t conv_tile;           /* 0: auto | 1: the 128-row conv_fwd_kernel with its full 128-column tiles (auto narrows them to 64
                                * columns for launches of at most half a tile per CU) | 2: the 256-row kernels wherever the shape allows
                                * | 3: the 512 x 128 form of conv_halo_kernel for any Cout that is a multiple of 128 (A/B of tile shapes) */
  int32_t conv_no_halo;        /* 1: conv_big_kernel where conv_halo_kernel would run (A/B, tests) */
  int32_t conv_big_min_tiles;  /* > 0: 256-row tiles a launch needs to go to the 256-row kernels (default 192) */
  int32_t max_workgroups;      /* > 0: cap on the persistent grids of conv_big / conv_halo (default: one per CU) */
  int32_t reserved_cus;        /* data-parallel runs: compute units the persistent kernels (256-row convs, wgrad_big,
                                * wgrad_halo) leave free — their workgroups own a CU for a whole tile loop, and RCCL's
                                * kernels (~130 latency-bound SyncBN all-reduces per step, the gradient buckets) then
                                * always find one.  0..128 */
  int32_t wgrad_kernel;        /* 0: auto | 1: wgrad_kernel (128 x 128 per-tap tiles) | 2: the wide kernels (wgrad_halo_kernel for
                                * 3x3 / stride 1, else wgrad_big_kernel) whenever the channel counts allow, whatever the
                                * pixel count (tests at small sizes) | 3: as 2 but never wgrad_halo_kernel (A/B timing) */
  int32_t wgrad_target_blocks; /* > 0: workgroups a weight-gradient launch aims for (split-K plan; A/B timing) */
  int32_t ablate;              /* tools/bench_conv.py: ablated variants of conv_fwd_kernel<128,128,64> — timing only */
  int32_t splitk_target_blocks; /* > 0: workgroups a split-K launch of the 128-row conv kernel aims for (default: three per
                                * four compute units; rn_conv_problem.splitk_ws; A/B timing) */
} rn_launch_opts;

/* Opaque per-device context: device id, compute-unit count, the default rn_launch_opts of the engine that owns it and
 * the RCCL communicators created through it (destroyed with it).  One per engine / Python process and GPU; not
 * thread-safe; functions are re-entrant across handles. */
typedef struct rn_handle rn_handle;
int rn_create(int device_id, rn_handle** out);
int rn_destroy(rn_handle* h);
int rn_handle_device(const rn_handle* h);
int rn_handle_num_cus(const rn_handle* h);
int rn_handle_set_launch_opts(rn_handle* h, const rn_launch_opts* opts);   /* RN_EINVAL on an out-of-range field */
int rn_handle_get_launch_opts(const rn_handle* h, rn_launch_opts* opts);
/* collective: rn_comm_init into slot `slot` (0..3) of the handle; rn_handle_comm returns it (NULL when unset) */
int rn_handle_comm_init(rn_handle* h, int slot, const void* unique_id /* host */, int rank, int world);
void* rn_handle_comm(const rn_handle* h, int slot);
/* destroys the communicator of `slot` (a no-op on an empty slot) and frees the slot: a bootstrap that failed on some rank
 * after rn_handle_comm_init abandons the communicator everywhere, and a later rn_handle_comm_init must find the slot empty */
int rn_handle_comm_destroy(rn_handle* h, int slot);

/* ---------------------------------------------------------------------------------------
 * K1/K2 (a5,a6,a8)  tf.keras.layers.Conv2D as used by resnet.py:118-144, fpn_base.py:44-50,
 * fpn.py:47-66, detection_head.py:56-88 — grouped implicit-GEMM convolution on MFMA.
 *
 * One launch runs up to RN_CONV_MAX_SEGMENTS independent problems that share kernel size,
 * stride and Cout tile shape (e.g. the five pyramid levels of one shared head conv, or the
 * FPN's per-level 3x3 convs).  Activations NHWC bf16; weights PREPACKED bf16
 * [Cout_pad][R][S][w_terms*Cin_pad] (rn_pack_conv_weight / rn_pack_conv_weight_split); accumulation fp32.
 *
 * Epilogue = the reference's layer sequence under the mixed_bfloat16 policy (__main__.py:76-77): every Keras
 * layer output (Conv2D incl. its bias, BatchNormalization, the residual `+`) is a bf16 TENSOR, so with
 * rb(v) = round-to-nearest-even to bf16:
 *   bf16 output:  t = acc + bias[c]                               (Conv2D + BiasAdd, fp32 inside the layer)
 *                 if scale/shift or residual given:  t = rb(t)    (the Conv2D layer's output tensor)
 *                 if scale/shift given:  t = t*scale[c] + shift[c];  if residual given: t = rb(t)   (BatchNorm output)
 *                 if residual given:     t = t + residual
 *                 y = rb(act(rb'(t)))    rb' = rb for swish (the Add / BatchNorm output feeds tf.nn.swish), identity else
 *   f32 output (the dtype=float32 prediction convs, detection_head.py:80-88): y = act((acc + bias)*scale + shift
 *                 + residual), no intermediate rounding.
 * w_terms (0/1 = plain): the f32 weights are carried as w_terms bf16 planes stacked along Cin
 * (w = w_hi + w_lo [+ w_lo2]); the kernel walks the same input channels once per plane, so a bf16 activation times
 * an f32 weight is accumulated to 16 (24) weight mantissa bits in fp32 — how the f32 prediction convs keep their
 * f32 kernels on a bf16 MFMA.
 * w_pair = 1 (f32 output only; instead of w_terms = 2): the two planes are stacked along COUT — packed rows
 * [64b, 64b + 32) = w_hi of channels [32b, 32b + 32), rows [64b + 32, 64b + 64) = w_lo of the same channels
 * (rn_pack_conv_weight_pair, rn_conv_pair_rows(Cout) rows) — the input is walked once, hi and lo products land in two
 * accumulator tiles of one wave and are added in the epilogue: y = (bias + sum x*w_hi) + sum x*w_lo.  For NARROW f32
 * layers (the 36-channel box prediction conv: 72 of 128 GEMM columns used, on the 512 x 128 tiles of the halo kernel,
 * instead of a 64-column tile of the 128-row kernel that stages the pixels once per tap and plane).  Only the 256- / 512-row
 * kernels take it: ask rn_conv_kernel_id() on the problem (0 -> RN_EINVAL at launch).
 */
#define RN_CONV_MAX_SEGMENTS 10

typedef struct {
  const void* x;        /* bf16 [N,H,W,pix_stride] (pix_stride >= Cin elements between pixels) */
  const void* w;        /* bf16 [Cout_pad, R, S, w_terms*Cin_pad]; w_pair: [rn_conv_pair_rows(Cout), R, S, Cin_pad] */
  void* y;              /* bf16 or f32 [N,Ho,Wo,Cout] */
  const float* scale;   /* f32[Cout] or NULL (=1) */
  const float* shift;   /* f32[Cout] or NULL (=0) */
  const void* residual; /* bf16 [N,Ho,Wo,Cout] or NULL */
  int32_t N, H, W, Cin, pix_stride, Ho, Wo, Cout;
  /* optional: BatchNorm forward statistics fused into the epilogue (bf16 outputs).  When this is non-NULL, every
   * 128-pixel row block b of the output writes its per-channel partial sums of the STORED bf16 values:
   * bn_partial[(b*2 + 0)*Cout + c] = sum, [(b*2 + 1)*Cout + c] = sum of squares — the stage-1 layout of
   * rn_bn_stats (rn_bn_segment.ext_chunks).  Blocks written: b = 0 .. 2*ceil(N*Ho*Wo/256)-1 when the launch runs on
   * the 256-row kernels (rn_conv_tile_rows() == 256; 4*ceil(N*Ho*Wo/512) blocks for 512: in general (rows / 128) *
   * ceil(N*Ho*Wo / rows)), b = 0 .. ceil(N*Ho*Wo/128)-1 on the 128-row kernel: set
   * ext_chunks accordingly. */
  float* bn_partial;
  const float* bias;    /* f32[Cout] or NULL: the Conv2D layer's bias, added to the fp32 accumulator */
  int32_t w_terms;      /* 0 or 1: plain bf16 weights; 2 / 3: split-bf16 planes along Cin (see above) */
  int32_t w_pair;       /* 1: two split-bf16 planes along Cout (see above; w_terms must be 0 / 1, f32 output) */
  /* optional: stage 1 of rn_bn_bwd_reduce fused into a DATA-GRADIENT launch.  The launch's output is dz, the gradient
   * that arrives at a BatchNorm + ReLU layer with no residual input (y = that layer's raw conv output, same shape as
   * this launch's output; bn_bwd_fwd = its rn_bn_segment.fwd: mean | invstd | scale | shift).  With both set (and
   * bn_partial; bf16 output, no scale / shift / bias / residual / activation on the launch itself), every 128-pixel
   * row block b writes   bn_partial[(b*2 + 0)*Cout + c] = sum g,   [(b*2 + 1)*Cout + c] = sum g*xhat,
   * g = dz_stored * [y*scale + shift > 0], xhat = (y - mean)*invstd — what stage 1 of rn_bn_bwd_reduce computes from
   * the same stored values (rn_bn_segment.ext_chunks_bwd).  Row blocks as for the forward statistics
   * (rn_conv_tile_rows()).  All segments of a launch or none. */
  const void* bn_bwd_y;
  const float* bn_bwd_fwd;
} rn_conv_segment;

typedef struct {
  int32_t R, S, stride_h, stride_w, pad_top, pad_left; /* x index = o*stride - pad + r */
  int32_t act;       /* RN_ACT_* applied after residual add */
  int32_t out_dtype; /* RN_DT_BF16 or RN_DT_F32 */
  int32_t num_segments;
  rn_conv_segment seg[RN_CONV_MAX_SEGMENTS];
  rn_launch_opts opts;
  /* optional split-K workspace (device memory, 16-byte aligned, ZERO-FILLED once by the caller; the kernels leave its
   * first 16 KB — the arrival counters — zero again after every launch).  The persistent 256-row kernels
   * walk their tiles in rounds of one tile per compute unit; the last round of a launch whose tile count is not a
   * multiple of the grid leaves most of the chip idle for a whole tile (8.3 rounds run as 9; a 100-tile launch uses 100
   * of 256 CUs).  With a workspace the halo kernel (3x3 / stride 1) runs the full rounds as before and the tiles of the
   * last round in a second launch, each cut along K (input-channel chunks, >= 4 per part, <= 4 parts) over several
   * workgroups: every part writes its fp32 accumulators here and counts itself in; the part that arrives LAST adds them
   * IN PART ORDER (deterministic: the same bits on every run, whoever arrives last) and runs the normal epilogue.  Nobody
   * waits for a partner: no spin, no timeout.  The 128-row kernel splits the same way (rn_conv.hip: small launches of
   * deep layers, every tile cut along K).  NULL / too small: whole tiles only.
   * rn_conv_splitk_workspace_bytes() = what this problem can use on the current device (0: it would not split).
   * Launches that share a workspace must be ordered on one stream.
   * Measured (round 4): for the last round of a BIG launch not faster — the 256 KB a part hands over cost what its half
   * tile of MFMA work saves — and the dispatcher does not do it (such launches run whole 512 x 128 tiles whenever that form
   * takes them).  For SMALL launches it pays: a 3x3 / stride 1 layer of fewer tiles than the 256-row kernels normally take
   * (ResNet stage 4 at batch 8: 26 tiles on 256 CUs) runs here with EVERY tile cut into up to four parts instead of on the
   * 128-row kernel: 68 -> 48 us.  The Python inference engine attaches a workspace by default, the two-stream training
   * engine under RNET_SPLITK=1. */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
} rn_conv_problem;

int rn_conv2d_nhwc_fwd(const rn_conv_problem* problem /* host */, void* stream);
size_t rn_conv_splitk_workspace_bytes(const rn_conv_problem* problem /* host */);
/* enough for any problem on any device (64 MB + 16 KB).  The dispatcher also looks at splitk_ws: a 3x3 / stride 1 launch
 * of fewer 256-row tiles than compute units (ResNet stage 3 / 4 at batch 8) goes to the halo kernel, every tile split,
 * only when a workspace is attached — set it BEFORE asking rn_conv_kernel_id / rn_conv_tile_rows. */
size_t rn_conv_splitk_workspace_max_bytes(void);

/* HWIO f32 [R,S,Cin,Cout] (the Keras kernel layout, resnet.py:137-144) -> bf16
 * [Cout_pad,R,S,Cin_pad], zero padded.  Cout_pad = rn_conv_cout_pad(Cout). */
int rn_conv_cout_pad(int Cout);
/* M-tile height the dispatcher picks for this problem: 256 = conv_big_kernel / conv_halo_kernel (256x256x32), 512 =
 * conv_halo_kernel's 512x128 form (3x3, 64 < Cout <= 128), 128 = conv_fwd_kernel<128,...>; 0 on a malformed problem.  A
 * launch with fused BatchNorm partial sums writes (rows / 128) * ceil(N*Ho*Wo / rows) 128-pixel row blocks per segment. */
int rn_conv_tile_rows(const rn_conv_problem* problem);
/* 128-pixel row blocks of fused BatchNorm partial sums (rn_conv_segment.bn_partial / bn_bwd_y) the launch writes for
 * `segment` — what rn_bn_segment.ext_chunks / ext_chunks_bwd must be set to.  (rows / 128) * ceil(N*Ho*Wo / rows) with
 * rows = rn_conv_tile_rows(), except for conv_big_kernel launches that run BALANCED tiles (HBM-bound single-segment 1x1
 * layers whose 256-row tiles would leave the last round of the persistent grid mostly idle are cut into tiles of fewer
 * rows, same number of rounds: two blocks per tile, the second one short).  0 on a malformed problem. */
int rn_conv_bn_row_blocks(const rn_conv_problem* problem, int segment);
/* Which kernel rn_conv2d_nhwc_fwd runs for `problem`: 0 = 128-row tiles (conv_fwd_kernel), 1 = conv_big_kernel
 * (256 x 256 x 32, persistent), 2 = conv_halo_kernel (256 x 256 x 32 for 3x3 / stride 1 / pad 1: pixels staged
 * once per channel chunk as a halo patch), 3 = conv_halo_kernel with 512 x 128 tiles (the same for 64 < Cout <= 128).
 * Profiling / bench bookkeeping only. */
int rn_conv_kernel_id(const rn_conv_problem* problem);
/* channel count of the packed weights: Cin rounded up to the kernel's K step (zero columns) */
int rn_conv_cin_pad(int Cin);
int rn_pack_conv_weight(const float* w_hwio, int R, int S, int Cin, int Cout, int Cin_pad, void* w_packed,
                        void* stream);
/* Split-bf16 packing for rn_conv_segment.w_terms = terms (2 or 3): plane 0 = rb(w), plane t = rb(w - sum of the
 * planes before it); bf16 [Cout_pad][R][S][terms][Cin_pad].  layout_ohwi = 0: w is HWIO (Keras), 1: [Cout][R][S][Cin]
 * (the training engine's master layout). */
int rn_pack_conv_weight_split(const float* w, int layout_ohwi, int R, int S, int Cin, int Cout, int Cin_pad, int terms,
                              void* w_packed, void* stream);
/* Packing for rn_conv_segment.w_pair: bf16 [rn_conv_pair_rows(Cout)][R][S][Cin_pad], rows as described there
 * (plane 0 = rb(w), plane 1 = rb(w - plane 0)); rn_conv_pair_rows(Cout) = 128 * ceil(Cout / 64). */
int rn_conv_pair_rows(int Cout);
int rn_pack_conv_weight_pair(const float* w, int layout_ohwi, int R, int S, int Cin, int Cout, int Cin_pad,
                             void* w_packed, void* stream);
/* Stem repack: 7x7x3 HWIO -> bf16 [64][7][32] rows = (kernel row r) x (8 taps x 4 channels),
 * tap 7 and channel 3 zero, matching rn_pack_stem_input's padded NHWC4 image. */
/* The same packing from the training engine's f32 master layout [Cout][R][S][Cin]. */
int rn_pack_conv_weight_ohwi(const float* w_ohwi, int R, int S, int Cin, int Cout, int Cin_pad, void* w_packed,
                             void* stream);
int rn_pack_stem_weight(const float* w_hwio, int Cout, void* w_packed, void* stream);
/* images f32 [N,H,W,3] -> bf16 [N,H+6,Wp,4] zero padded by 3 (fixed_padding, resnet.py:92-115),
 * Wp = rn_stem_padded_width(W). */
int rn_stem_padded_width(int W);
int rn_pack_stem_input(const float* images, int N, int H, int W, void* packed, void* stream);
/* General form of the two calls above for any first-layer conv on 3-channel images whose kernel
 * width S <= 8 and stride is 2 (EfficientNet Stem 3x3 s2 SAME, efficientnet.py:566-586): weights
 * HWIO [R,S,3,Cout] -> bf16 [Cout_pad][R][8][4]; images f32 [N,H,W,3] -> bf16 [N,Hp,Wp,4] with
 * pad_top/pad_left zero rows/cols in front, zero fill behind (Wp % 8 == 0) and a zero 4th channel.
 * The conv is then rn_conv2d_nhwc_fwd with R=R, S=1, Cin=32, pix_stride=4, stride 2, pad 0. */
int rn_pack_stem_weight_rs(const float* w_hwio, int R, int S, int Cout, void* w_packed, void* stream);
int rn_pack_image_nhwc4(const float* images, int N, int H, int W, int pad_top, int pad_left, int Hp, int Wp,
                        void* packed, void* stream);

/* ---------------------------------------------------------------------------------------
 * Backward of K1/K2 (the tape.gradient of executor.py:427-428 through Conv2D).
 * Weight gradient: dw f32 [Cout][R][S][Cin] (the packed compute layout) = beta*dw + sum over
 * all segments' output pixels of dy[p][co] * x[shifted pixel][ci].  Segments share one weight
 * (pyramid levels of a shared head conv).  x bf16 [N,H,W,Cin], dy bf16 [N,Ho,Wo,Cout].
 * Deterministic split-K: partial tiles go to the workspace and are added in index order.
 */
typedef struct {
  const void* x;
  const void* dy;
  int32_t N, H, W, Cin, Ho, Wo, Cout;
  int32_t dy_pix_stride; /* elements between dy pixels; 0 = Cout (dense) */
  int32_t x_pix_stride;  /* elements between x pixels; 0 = Cin (dense); 4 for the packed stem image */
} rn_wgrad_segment;

typedef struct {
  int32_t R, S, stride_h, stride_w, pad_top, pad_left;
  int32_t num_segments;
  rn_wgrad_segment seg[RN_CONV_MAX_SEGMENTS];
  rn_launch_opts opts;
} rn_wgrad_problem;

size_t rn_wgrad_workspace_bytes(const rn_wgrad_problem* problem /* host */);
/* Which kernel rn_conv2d_nhwc_wgrad runs for `problem`: 0 = wgrad_kernel (128 x 128 per-tap tiles, K step 64 pixels),
 * 1 = wgrad_big_kernel (256 x 256 per-tap tiles, ping-pong), 2 = wgrad_halo_kernel (3x3 / stride 1 / pad 1: all nine taps
 * in one workgroup, reduction over image rows); -1 on a malformed problem.  Bench bookkeeping only. */
int rn_wgrad_kernel_id(const rn_wgrad_problem* problem /* host */);
int rn_conv2d_nhwc_wgrad(const rn_wgrad_problem* problem /* host */, float* dw, float beta, void* workspace,
                         size_t workspace_bytes, void* stream);
/* n layers' weight gradients as one call: the same results as n calls of rn_conv2d_nhwc_wgrad (dws[i] for problems[i], both
 * host arrays).  Layers of IDENTICAL geometry that wgrad_halo_kernel serves (n <= 8: the eight head-tower layers, the 3x3
 * layers of one ResNet stage) run as ONE launch over (layer, co tile, ci tile) tiles + one reduction launch: a split-K
 * workgroup writes its whole 288 KB accumulator as a partial tile, so a launch costs ~75 MB of partials however small
 * the layer, and a grouped launch needs 1/n of the pixel chunks per layer.  rn_wgrad_group_fused tells which way a group
 * goes (1: one launch); other groups are issued layer by layer.  Deterministic either way. */
size_t rn_wgrad_group_workspace_bytes(const rn_wgrad_problem* const* problems /* host */, int n);
int rn_wgrad_group_fused(const rn_wgrad_problem* const* problems /* host */, int n);
int rn_conv2d_nhwc_wgrad_group(const rn_wgrad_problem* const* problems /* host */, int n, float* const* dws /* host */,
                               float beta, void* workspace, size_t workspace_bytes, void* stream);

/* dgrad: the data gradient of a stride-1 conv is rn_conv2d_nhwc_fwd run on dy with these weights
 * (bf16 [Cin_pad][R][S][Cout_pad], taps flipped, zero rows/columns in the padding) and pad' = k-1-pad; stride 2 goes through
 * rn_upsample_zero2x first.  w_ohwi is the f32 master in compute layout [Cout][R][S][Cin]. */
int rn_pack_conv_weight_dgrad(const float* w_ohwi, int R, int S, int Cin, int Cout, int Cout_pad, void* w_packed,
                              void* stream);
/* the same for many layers in one launch (host array of descriptors; any n, chunked by RN_DGRAD_PACK_MAX) */
#define RN_DGRAD_PACK_MAX 64
typedef struct {
  const float* w_ohwi; /* f32 master [Cout][R][S][Cin] */
  void* w_packed;      /* bf16 [cout_pad(Cin)][R][S][Cout_pad];  pad_ == 1: bf16 [cout_pad(4*Cin)][2][2][Cout_pad] */
  int32_t R, S, Cin, Cout, Cout_pad;
  int32_t pad_;        /* 0: the form above.  1: sub-pixel form of a 3x3 / stride 2 / pad 1 layer (even input size): the data
                        * gradient is rn_conv2d_nhwc_fwd on dy with R = S = 2, stride 1, pad 0, output size = dy's size,
                        * 4*Cin output channels (phase (a,b) of dx[2i+a][2j+b] in channels (a*2+b)*Cin ..), followed by
                        * rn_depth_to_space2x: 16 tap products per dy pixel instead of 36 for the zero-upsampled form. */
} rn_dgrad_pack;
int rn_pack_conv_weight_dgrad_batch(const rn_dgrad_pack* items /* host */, int n, void* stream);
/* f32 [P,C] -> bf16 [P,Cpad], zero padded channels (dy of the 36/720-channel prediction convs is
 * padded to a multiple of 64 so it can be the K dimension of the dgrad GEMM) */
int rn_cast_pad_f32_to_bf16(const float* x, void* y, int64_t P, int C, int Cpad, void* stream);
/* dst[c] = add + sum_{r < rows} src[r * row_stride + c] for c < n, rows added in index order: the bias gradient of a conv
 * shared by the pyramid levels (sum of its per-level column sums, executor.py:427-428 through tape.gradient) and the
 * batch's positive count + 1 (retinanet_loss.py:38) without a library reduction. */
int rn_reduce_rows_f32(const float* src, int rows, int64_t row_stride, int n, float add, float* dst, void* stream);
/* y[n,2h,2w,:] = x[n,h,w,:], zero elsewhere; bf16 NHWC */
int rn_upsample_zero2x(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
/* y[n,2h,2w,:] (+)= x[n,h,w,:], bf16: accumulate = 0 is rn_upsample_zero2x (zeros at the other positions),
 * accumulate = 1 adds in fp32 and rounds once, touching only the even positions.  Data gradient of a 1x1 / stride-2
 * convolution (the ResNet projection shortcuts): the GEMM runs on dy at the low resolution. */
int rn_scatter_add2x(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int accumulate, void* stream);
/* y[n, 2i+a, 2j+b, c] (+)= x[n, i, j, (a*2+b)*C + c]; x bf16 [N,H,W,4C], y bf16 [N,2H,2W,C] */
int rn_depth_to_space2x(const void* x, void* y, int N, int H, int W, int C, int accumulate, void* stream);
int rn_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
/* dy = dz * act'(z) for a layer with an activation but no BatchNorm in front */
int rn_act_bwd(const void* dz, const void* z, void* dy, int64_t n, int act, void* stream);

/* ---------------------------------------------------------------------------------------
 * K3  BatchNormalization / SyncBatchNormalization in training mode (model/utils.py:7-22, used at
 * resnet.py:59-79, fpn.py:54-69, detection_head.py:68-74,99), forward and backward, with the
 * activation and the residual add of resnet.py:248 fused.  Grouped like the convs.
 *   rn_bn_stats        y -> sums[2][C] = (sum, sum of squares) of THIS rank's pixels
 *   (SyncBN: the caller all-reduces `sums` over ranks and sets count_scale = replicas)
 *   rn_bn_finalize     sums -> fwd[4][C] = (mean, invstd, scale, shift); moving stats update
 *   rn_bn_apply        z = act((y*scale + shift) [* sample_scale[n]] + residual); the BatchNorm output, the
 *                      drop_connect output and (in front of swish) the residual sum are rounded to bf16 where the
 *                      reference holds a bf16 tensor between two layers (see rn_conv_segment)
 *   rn_bn_bwd_reduce   dz, z, y -> bsums[2][C] = (sum g, sum g*xhat), g = dz*act'(z); also dbeta = bsums[0],
 *                      dgamma = bsums[1]: the gamma / beta gradients are THIS replica's sums
 *   (SyncBN: all-reduce `bsums` — after rn_bn_bwd_reduce, so dgamma / dbeta stay local like tf.gradients leaves them
 *    until the optimizer's cross-replica sum)
 *   rn_bn_bwd_apply    dy = scale*(g - sum_g/n - xhat*sum_gxhat/n); dres (+)= g
 * All tensors bf16 [P,C] (P = N*H*W), per-channel arrays f32.
 */
typedef struct {
  const void* y;
  void* z;
  const void* residual;
  const void* dz;
  void* dy;
  void* dres;
  float* sums;
  float* fwd;
  float* bsums;
  const float* gamma;
  const float* beta;
  float* moving_mean;
  float* moving_var;
  float* dgamma;
  float* dbeta;
  int64_t P;
  int32_t C;
  int32_t dres_accumulate;
  /* drop_connect / stochastic depth (backbone/efficientnet.py:97-113): optional f32 [P / rows_per_sample]
   * per-image factors (0 or 1/survival_prob) applied to the BatchNorm output BEFORE the residual add:
   * z = act((y*scale + shift) * m[n] + residual); the backward scales the BN branch by m[n], dres is not. */
  const float* sample_scale;
  int64_t rows_per_sample;
  /* > 0: the stage-1 partial sums of rn_bn_stats were already written by the producing convolution
   * (rn_conv_segment.bn_partial) as `ext_chunks` row blocks at workspace + rn_bn_partial_offset_bytes(); rn_bn_stats
   * then only runs the ordered final reduction.  All segments of a problem must agree. */
  int32_t ext_chunks;
  /* > 0: stage 1 of rn_bn_bwd_reduce was written by the data-gradient launch that produced dz
   * (rn_conv_segment.bn_bwd_y) as `ext_chunks_bwd` row blocks of (sum g, sum g*xhat) at workspace +
   * rn_bn_bwd_partial_offset_bytes(); rn_bn_bwd_reduce then only runs the ordered final reduction.  Only for
   * act = relu without residual and without sample_scale.  All segments or none. */
  int32_t ext_chunks_bwd;
  /* optional, u8 [P*C/8]: gate bits of a relu / relu6 that follows a residual add.  rn_bn_apply writes bit (c & 7) of
   * byte (p*C + c)/8 = [act'(z) != 0] for the bf16 z it stores; with it set on every segment rn_bn_bwd_reduce and
   * rn_bn_bwd_apply read these P*C/8 bytes instead of z (2*P*C bytes) — same gate, same results. */
  void* act_mask;
  /* optional, f32: rn_bn_bwd_apply also writes the column sums of the bf16 dy it STORES, per workgroup chunk, in the
   * stage-1 layout of rn_bn_stats — row (chunk*2 + 0)*C + c, rn_bn_bwd_colsum_chunks() chunks (the second row of each
   * chunk is not touched).  The bias of a Conv2D in front of this BatchNorm has the gradient sum_p dy[p][c]
   * (tape.gradient, executor.py:427-428): with this the engine finishes it with rn_bn_stats(ext_chunks) on the partials
   * instead of reading the whole dy tensor again.  Same values as summing the stored tensor, up to fp32 association. */
  float* dy_colsum_partial;
} rn_bn_segment;

typedef struct {
  int32_t num_segments;
  int32_t act;
  int32_t bessel;        /* 1: moving variance uses the n/(n-1) corrected batch variance (fused BN) */
  float eps;
  float momentum;
  float count_scale;     /* replicas contributing to `sums` (1 for local BN) */
  rn_bn_segment seg[RN_CONV_MAX_SEGMENTS];
} rn_bn_problem;

/* Workspace of the reduction passes: [per-chunk partial sums][ticket counters][part slots].  A segment with more than
 * 512 rows of partial sums has its stage-2 reduction cut into parts whose last arriver (an atomic ticket) adds the parts
 * in part order; the ticket counters must be ZERO before the first launch on a workspace — zero the workspace once after
 * allocating it — and every launch leaves them zero (the contract of rn_conv_problem.splitk_ws).
 * Ordering: the counters are shared by the forward (rn_bn_stats*) and the backward (rn_bn_bwd_reduce) launches on one
 * workspace, so all launches that use one workspace must be ordered on ONE stream (or by events). */
size_t rn_bn_workspace_bytes(const rn_bn_problem* problem /* host */);
/* zeroes the ticket counters only (a few hundred bytes, one asynchronous memset on `stream`): for a workspace that was
 * not zero-filled at allocation.  RN_ENOMEM when the workspace is smaller than rn_bn_workspace_bytes(problem). */
int rn_bn_workspace_init(const rn_bn_problem* problem, void* workspace, size_t workspace_bytes, void* stream);
size_t rn_bn_partial_offset_bytes(const rn_bn_problem* problem, int segment);   /* forward-stats partials in the workspace */
size_t rn_bn_bwd_partial_offset_bytes(const rn_bn_problem* problem, int segment);   /* backward partials (ext_chunks_bwd) */
int rn_bn_stats(const rn_bn_problem* problem, void* workspace, size_t workspace_bytes, void* stream);
/* rn_bn_stats + rn_bn_finalize in one call for single-replica BatchNorm (no all-reduce between them): the final
 * reduction kernel finishes each channel with the finalize arithmetic — same values, one launch less per layer */
int rn_bn_stats_finalize(const rn_bn_problem* problem, void* workspace, size_t workspace_bytes, void* stream);
int rn_bn_finalize(const rn_bn_problem* problem, void* stream);
int rn_bn_apply(const rn_bn_problem* problem, void* stream);
int rn_bn_bwd_reduce(const rn_bn_problem* problem, void* workspace, size_t workspace_bytes, void* stream);
int rn_bn_bwd_apply(const rn_bn_problem* problem, void* stream);
/* chunks rn_bn_bwd_apply writes per segment when rn_bn_segment.dy_colsum_partial is set (ceil(P * C/8 / 2048): one per
 * workgroup of the chunked elementwise form); 0 = this problem runs the grid-stride form (a segment's C/8 does not divide
 * 256), which does not support it — leave dy_colsum_partial NULL then. */
int rn_bn_bwd_colsum_chunks(const rn_bn_problem* problem, int segment);

/* backward of K6/K7/K8 */
int rn_maxpool2d_nhwc_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int k, int stride,
                          int pad_top, int pad_left, int Ho, int Wo, int accumulate, void* stream);
/* one level of the FPN top-down backward: din = (dout + sum2x2(din_finer)) * act'(out) */
int rn_fpn_topdown_bwd_level(const void* dout, const void* din_finer, const void* out, void* din, int N, int H,
                             int W, int C, int act, void* stream);
/* davg_scratch: rn_balance_features_bwd_scratch_bytes() bytes — d_avg at the intermediate level, then one byte per (pixel,
 * channel) of every coarser level: where in its pooling window the average has its first maximum, found once per coarse
 * pixel instead of by every fine pixel of the window.  RN_ENOMEM when too small. */
size_t rn_balance_features_bwd_scratch_bytes(int num_levels, int mid, int N, int H0, int W0, int C);
int rn_balance_features_bwd(void* const* dout, void* const* in, void* const* din, const void* avg,
                            void* davg_scratch, size_t scratch_bytes, int num_levels, int mid, int N, int H0, int W0,
                            int C, void* stream);

/* ---------------------------------------------------------------------------------------
 * a10  Executor._train_step's optimizer side (executor.py:296-327,401-441; optimizers/builder.py:27-71)
 * over flat f32 arenas.  segs_dev: device array of {i64 offset, i64 size, i32 weight_decay,
 * i32 block_begin, i32 nblocks, i32 pad, i64 bf16_offset}; block_seg_dev: device i32[num_blocks]
 * (block -> segment), blocks of rn_optim_chunk() elements.
 *   rn_optim_clip      g *= grad_unscale (LossScaleOptimizer.get_unscaled_gradients, executor.py:429-430; 1 = off);
 *                      g += wd_coeff*w (decayed tensors); per-tensor clip_by_norm then clip_by_global_norm;
 *                      metrics f32[8] = {norm after, norm before, global factor, l2-regularization =
 *                      wd_alpha * sum ||w||^2 / 2 over the decayed tensors (executor.py:296-299), 1 if any clip
 *                      factor != 1, 1 if the gradient norm is not finite, -, -}
 *   rn_optim_clip_prepare / _factors / _apply: the three stages on their own.  The data-parallel step launches the
 *                      gradient all-reduce bucket by bucket while the backward pass is still running, on UNclipped
 *                      gradients (the clip factors need every gradient): `prepare` runs per bucket (blocks
 *                      [block_begin, block_begin + block_count)), `factors` once at the end and also writes
 *                      flags[0] = "a factor != 1 on this rank, or its gradient norm is not finite", flags[1] = "not finite" (slots that ride in the
 *                      last bucket's all-reduce); if some rank's clip fired, `apply` with correction != NULL writes
 *                      (factor - 1) * g, which is all-reduced and added — the sum then equals the reference's
 *                      clip-then-all-reduce order (executor.py:432-437).
 *   rn_optim_sgd_step  v = m v - lr g; w += v (or w += m v - lr g with the new v: nesterov); ema -= (1-d)(ema - w);
 *                      bf16 copy of w.  skip_flag (device f32, may be NULL): non-zero = drop the step
 *                      (LossScaleOptimizer.apply_gradients with non-finite gradients).
 */
int rn_optim_chunk(void);
size_t rn_optim_workspace_bytes(int num_blocks, int num_segments);
int rn_optim_clip(float* grads, const float* params, const void* segs_dev, int num_segments,
                  const int32_t* block_seg_dev, int num_blocks, float wd_coeff, float wd_alpha, float grad_unscale,
                  float clipnorm, float* metrics, void* workspace, size_t workspace_bytes, void* stream);
int rn_optim_clip_prepare(float* grads, const float* params, const void* segs_dev, const int32_t* block_seg_dev,
                          int num_blocks, int block_begin, int block_count, float wd_coeff, float grad_unscale,
                          float* local_copy /* same layout as grads, or NULL: keeps this rank's gradients */,
                          void* workspace, size_t workspace_bytes, void* stream);
int rn_optim_clip_factors(const void* segs_dev, int num_segments, int num_blocks, float clipnorm, float wd_alpha,
                          float* metrics, float* flags, void* workspace, size_t workspace_bytes, void* stream);
int rn_optim_clip_apply(float* grads, float* correction, const void* segs_dev, const int32_t* block_seg_dev,
                        int num_blocks, void* workspace, size_t workspace_bytes, void* stream);
int rn_optim_sgd_step(float* params, const float* grads, float* momentum_buf, float* ema, void* params_bf16,
                      const void* segs_dev, const int32_t* block_seg_dev, int num_blocks, float lr, float momentum,
                      float ema_decay, int nesterov, const float* skip_flag, void* stream);

/* ---------------------------------------------------------------------------------------
 * C1-C3  collectives of the data-parallel step over RCCL / xGMI (the reference reaches them through
 * tf.distribute: retinanet_loss.py:46-49, model/utils.py:10-12, executor.py:436-437).  One process per GPU; rank 0
 * draws a unique id (rn_comm_unique_id, rn_comm_unique_id_bytes() bytes) and hands it to every rank out of band;
 * rn_comm_init is collective: agree on rn_comm_available() over the job before any rank calls it, and on its status
 * afterwards before any rank uses (or abandons) the communicator.  A collective is enqueued on the CALLER'S stream (asynchronous to the host, in stream
 * order with the kernels around it); in-place SUM.  Use one communicator per stream that carries collectives.
 * rn_allreduce_bucket: gradient buckets (RN_DT_F32 / RN_DT_BF16); rn_allreduce_small: the few-KB fp32 messages
 * (SyncBatchNorm [sum | sum of squares], the loss normaliser) on the latency path.  Status RN_ECOMM on failure. */
int rn_comm_available(void);   /* local check, no collective: 1 if librccl loads with every entry point rn_comm needs */
int rn_comm_unique_id_bytes(void);
int rn_comm_unique_id(void* out /* host, rn_comm_unique_id_bytes() bytes */);
int rn_comm_init(const void* unique_id /* host */, int rank, int world, void** comm_out);
int rn_comm_destroy(void* comm);
int rn_allreduce_bucket(void* comm, void* ptr /* device */, int64_t count, int dtype, void* stream);
int rn_allreduce_small(void* comm, float* ptr /* device */, int count, void* stream);

/* ---------------------------------------------------------------------------------------
 * Measurement infrastructure for SURVEY 8(d) (bench.py's roofline): the matrix-pipe rate this part SUSTAINS.
 * rn_probe_mfma runs only v_mfma_f32_32x32x16 (8 waves per workgroup = 2 per SIMD, one resident workgroup per CU, 4 x CUs
 * workgroups, `iters` x 16 MFMAs per wave; no memory traffic) on operands converted from `table` (device f32[1024]: random
 * values or zeros — the chip clocks to its power budget, so the rate depends on the data).  out: device f32[4*CUs*512]
 * (keeps the work live).  clocks: device u64[4] = shader clock (s_memtime) at the first / last instruction of workgroup 0,
 * 100 MHz wall clock at the same points: (c1 - c0) / ((w1 - w0) / 100) = core clock in MHz.  rn_probe_mfma_flops(iters) =
 * FLOPs of one launch on the current device. */
long long rn_probe_mfma_flops(int iters);
/* One wavefront that idles for `microseconds` (0..100000) on `stream`: the probe behind the engines' choice of side streams
 * — HIP maps streams onto a few hardware queues by creation order, and two streams on one queue do not overlap (a kernel
 * launched on stream B after a spin on stream A finishes first only when A and B run side by side). */
int rn_probe_spin(int microseconds, void* stream);
int rn_probe_mfma(const float* table, float* out, int iters, unsigned long long* clocks, void* stream);

/* ---------------------------------------------------------------------------------------
 * K6  tf.keras.layers.MaxPool2D  (resnet.py:304-307 3x3 s2 SAME; fpn_base.py:25-26,68 2x2 s2)
 * explicit pads; padded taps are skipped (-inf). bf16 NHWC.
 */
int rn_maxpool2d_nhwc(const void* x, void* y, int N, int H, int W, int C, int k, int stride, int pad_top,
                      int pad_left, int Ho, int Wo, void* stream);

/* K1 + K6 fused for the ResNet stem with BatchNorm in inference form (`resnet_initial` frozen, builder.py:28-29;
 * serving): Conv2D 7x7/2 fixed padding + scale/shift + relu|relu6 + MaxPool 3x3/2 (resnet.py:288-307) in one launch —
 * the stem output never goes to HBM.  x: bf16 [N,Hp,Wp,4] from rn_pack_image_nhwc4; w_packed: rn_pack_stem_weight_rs
 * (R = S = 7, Cout = 64); Hs x Ws: the conv output size; y: bf16 [N,Po,Qo,64].  Same arithmetic and rounding points
 * as rn_conv2d_nhwc_fwd (R=7, S=1, Cin=32, pix_stride=4, stride 2) followed by rn_maxpool2d_nhwc: bit-identical.
 * RN_EINVAL for any other layer shape (callers keep the two-launch path). */
int rn_stem_conv_bn_relu_pool(const void* x, const void* w_packed, const float* scale, const float* shift, void* y,
                              int N, int Hp, int Wp, int Hs, int Ws, int R, int Cout, int act, int pool_k,
                              int pool_stride, int pool_pad_top, int pool_pad_left, int Po, int Qo, void* stream);

/* a5, cross-layer: ONE launch per bottleneck block of ResNet stage 1 with every BatchNorm in inference form (`resnet_initial`
 * frozen, builder.py:28-29 — the layers executor.py:154-176 runs with training=False; every layer when serving):
 * retinanet/model/backbone/resnet.py:194-248 `bottleneck_block` with filters = 64, strides = 1 (block_group1, :324-331)
 *     a   = relu(BN(conv1x1 Cx -> 64 (x)));  b = relu(BN(conv3x3 64 -> 64 (a)))  (zero padding 1);
 *     out = relu(BN(conv1x1 64 -> 256 (b)) + shortcut),  shortcut = x (Cx = 256) | BN(conv1x1 64 -> 256 (x)) (Cx = 64: the
 *     projection shortcut of the group's first block, :220-228).
 * a and b never go to HBM (rn_bneck.hip).  Rounding points = those of three (four) rn_conv2d_nhwc_fwd launches (Conv2D output
 * -> bf16, BatchNorm -> bf16, + shortcut, relu -> bf16); the K orders differ, so results agree to fp32 summation order.
 *   x: bf16 [N,H,W,Cx] contiguous; y: bf16 [N,H,W,256]; W % 32 == 0 and W <= 160 (rn_bottleneck64_supported; other shapes:
 *   RN_EINVAL, callers keep the per-layer launches);
 *   w_packed: rn_bottleneck64_pack of the layers' f32 HWIO kernels (Keras layout) = MFMA fragments [wa | wb | wo | wsc],
 *   rn_bottleneck64_packed_bytes(Cx) bytes, 16-byte aligned; wsc_hwio = NULL iff Cx == 256;
 *   affine: f32 [a_scale 64][a_shift 64][b_scale 64][b_shift 64][o_scale 256][o_shift 256] (+ [sc_scale 256][sc_shift 256]
 *   when Cx == 64): BatchNorm folded to scale = gamma / sqrt(var + eps), shift = beta - mean * scale. */
typedef struct {
  const void* x;
  void* y;
  const void* w_packed;
  const float* affine;
  int32_t N, H, W, Cx;
  rn_launch_opts opts;   /* reserved_cus / max_workgroups bound the grid (one workgroup per compute unit) */
} rn_bottleneck64_problem;
int rn_bottleneck64_supported(int N, int H, int W, int Cx);   /* 1 | 0, host only */
size_t rn_bottleneck64_packed_bytes(int Cx);
int rn_bottleneck64_pack(const float* wa_hwio, const float* wb_hwio, const float* wo_hwio, const float* wsc_hwio, int Cx,
                         void* packed, void* stream);
int rn_bottleneck64_fwd(const rn_bottleneck64_problem* problem, void* stream);

/* K7 (a6)  FPN top-down path, FeatureFusion mode 'sum' + NearestUpsampling2D + activation
 * (fpn.py:93-98, feature_fusion.py:41-56, nearest_upsampling.py:19-21):
 * for l = num_levels-1 .. 1: out[l-1] = act(in[l-1] + up2(out[l])), out[top] = in[top] (not
 * written); p[l] bf16 [N,H0>>l,W0>>l,C] with level 0 the finest; out must not alias in. */
int rn_fpn_topdown(void* const* p_in /* host array of device ptrs */, void* const* p_out, int num_levels,
                   int N, int H0, int W0, int C, int act, void* stream);

/* K8 (a7)  BalanceFeatures  (balance_features.py:19-60); out[l] may alias in[l];
 * mid = index of the intermediate level; scratch: bf16 [N,Hmid,Wmid,C]. */
int rn_balance_features(void* const* p_in, void* const* p_out, int num_levels, int mid, int N, int H0, int W0,
                        int C, void* scratch, void* stream);

/* ---------------------------------------------------------------------------------------
 * a18  EfficientNet / SeparableConv2D pieces that are not GEMMs
 * (retinanet/model/backbone/efficientnet.py:222-265 SE, :372-380 depthwise conv;
 *  model/neck/fpn_base.py:28-39 and model/head/detection_head.py:37-50 SeparableConv2D)
 * Depthwise k x k conv, TF SAME pads given explicitly, y = act(dw(x)*scale[c] + shift[c]), bf16 NHWC,
 * weights bf16 [k*k][C] (rn_pack_depthwise_weight from the Keras [k,k,C,1] f32 kernel).  Grouped.
 */
typedef struct {
  const void* x;
  const void* w;
  void* y;
  const float* scale;
  const float* shift;
  const void* residual;  /* optional bf16 [N,Ho,Wo,C] added before the activation (gradient accumulation) */
  int32_t N, H, W, C, Ho, Wo;
} rn_dw_segment;

typedef struct {
  int32_t k, stride, pad_top, pad_left, act, num_segments;
  rn_dw_segment seg[RN_CONV_MAX_SEGMENTS];
} rn_dw_problem;

int rn_depthwise_conv2d_nhwc_fwd(const rn_dw_problem* problem /* host */, void* stream);
int rn_pack_depthwise_weight(const float* w_kkc1, int k, int C, void* w_packed, void* stream);
/* Backward of the depthwise conv (tape.gradient through DepthwiseConv2D / SeparableConv2D, executor.py:427-428):
 *   data gradient   = rn_depthwise_conv2d_nhwc_fwd on dy (zero-upsampled with rn_upsample_zero2x for stride 2)
 *                     with the tap-reversed filter (rn_pack_depthwise_weight_flip from the f32 master [k*k][C]),
 *                     stride 1, pads k-1-pad; `residual` accumulates into an already written gradient buffer;
 *   weight gradient = rn_depthwise_conv2d_nhwc_wgrad: problem.seg[i].x = layer input, .y = dy (read only);
 *                     dw f32 [k*k][C] (the Keras [k,k,C,1] order) summed over all segments; k in {1,3,5};
 *                     deterministic two-stage reduction through `workspace`. */
int rn_pack_depthwise_weight_flip(const float* w_kkc, int k, int C, void* w_packed, void* stream);
size_t rn_depthwise_wgrad_workspace_bytes(const rn_dw_problem* problem);
int rn_depthwise_conv2d_nhwc_wgrad(const rn_dw_problem* problem, float* dw, void* workspace, size_t workspace_bytes,
                                   void* stream);
/* SE in place on x bf16 [N,HW,C]: x *= sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2).
 * w_reduce bf16 [se][C], w_expand bf16 [C][se], biases f32. */
size_t rn_se_workspace_bytes(int N, int C);
int rn_squeeze_excite_inplace(void* x, int N, int HW, int C, const void* w_reduce, const float* b_reduce,
                              const void* w_expand, const float* b_expand, int se, void* workspace,
                              size_t workspace_bytes, void* stream);
/* Training forward (out of place; `state`, rn_se_workspace_bytes(N,C) bytes, keeps pooled / gate / h1 / a) and
 * backward: dx = dy*gate + d(pooled)/HW (dx may alias dy), parameter gradients dw1 f32 [se][C], db1 [se],
 * dw2 [C][se], db2 [C] overwritten. */
int rn_squeeze_excite_fwd(const void* x, void* y, int N, int HW, int C, const void* w_reduce, const float* b_reduce,
                          const void* w_expand, const float* b_expand, int se, void* state, size_t state_bytes,
                          void* stream);
int rn_squeeze_excite_bwd(const void* x, const void* dy, void* dx, int N, int HW, int C, const void* w_reduce,
                          const void* w_expand, int se, const void* state, float* dw1, float* db1, float* dw2,
                          float* db2, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * §8(f)-1  prepare_image / validation preprocessing
 * (retinanet/dataloader/preprocessing_pipeline.py:96-121, retinanet/dataloader/utils.py:58-66)
 * image f32[h,w,3] (RGB, raw pixel values) -> out f32[target_h,target_w,3]:
 * (x/pixel_scale - mean)/stddev, TF2 bilinear resize to [scaled_h,scaled_w] (half-pixel centres),
 * zero padding bottom/right.  scaled_h/w = round(shape * min(target/shape)) is computed by the
 * caller exactly as the reference does in float32 (:98-101).  mean/stddev: host f32[3].
 */
int rn_prepare_image(const float* image, int h, int w, int scaled_h, int scaled_w, float* out, int target_h,
                     int target_w, const float* mean, const float* stddev, float pixel_scale, void* stream);

/* ---------------------------------------------------------------------------------------
 * §8(f)-2  COCOEvaluator.accumulate_results  (retinanet/eval/coco_evaluator.py:95-134)
 * boxes f32[B,D,4] (x1,y1,x2,y2 as produced by the serving signature), classes i32[B,D], valid i32[B],
 * resize_scale f32[B,2] (the `resize_scale` of preprocessing_pipeline.py:96-121) ->
 * out_bbox i32[B,D,4] = (x, y, w, h) after `boxes /= tile(resize_scale / input_shape, 2)` and int32 truncation,
 * out_category i32[B,D] = class_lut[class] (sorted-name -> original COCO id, :39-52, 89-93) or the class itself
 * when class_lut is NULL; slots >= valid[b] get bbox 0 and category -1.
 */
int rn_coco_accumulate(const float* boxes, const int32_t* classes, const int32_t* valid, const float* resize_scale,
                       float input_h, float input_w, const int32_t* class_lut, int num_classes, int B, int D,
                       int rescale, int32_t* out_bbox, int32_t* out_category, void* stream);

/* ---------------------------------------------------------------------------------------
 * §8(f)-4  TFRecord input format — HOST functions (every pointer below is host memory; no HIP call)
 * Replaces what tf.data.TFRecordDataset (retinanet/dataloader/input_pipeline.py:60-68) and
 * tf.io.parse_single_example (retinanet/dataloader/tfrecord_parser.py:4-41) do for the reference, and
 * tf.io.TFRecordWriter / tf.train.Example.SerializeToString (retinanet/dataset_utils/tfrecord_writer.py:27-57).
 *
 * Framing of one record: u64le length | u32le masked_crc32c(length bytes) | payload | u32le masked_crc32c(payload),
 * masked = rotr(crc, 15) + 0xa282ead8, crc = CRC-32C (Castagnoli, reflected 0x82F63B78).
 */
uint32_t rn_crc32c(const void* data, size_t nbytes);
uint32_t rn_crc32c_masked(const void* data, size_t nbytes);
/* Walk a buffer of whole records.  Fills payload_offsets/lengths (relative to buf) for at most max_records
 * records, returns how many were found (>= 0) and the bytes they cover in *consumed (a truncated trailing
 * record is not an error when allow_partial_tail != 0: it is left unconsumed, like a reader that refills its
 * buffer).  verify_crc != 0 checks both checksums.  Negative rn_status on corruption (TF: DataLossError). */
long long rn_tfrecord_scan(const uint8_t* buf, size_t nbytes, uint64_t* payload_offsets, uint64_t* payload_lengths,
                           long long max_records, int verify_crc, int allow_partial_tail, size_t* consumed);
/* Frame one payload into out (capacity >= nbytes + 16); returns the bytes written. */
size_t rn_tfrecord_frame(const uint8_t* payload, size_t nbytes, uint8_t* out);

/* One parsed tf.train.Example with the reference's feature set (tfrecord_parser.py:5-13). */
typedef struct {
  uint64_t image_offset, image_length; /* the 'image' bytes inside the record (FixedLenFeature [], string) */
  int64_t image_id;                    /* 'image_id' (FixedLenFeature [], int64) */
  int32_t n_xmins, n_ymins, n_xmaxs, n_ymaxs, n_classes; /* VarLenFeature lengths (0 when absent) */
  int32_t pad_;
} rn_example_info;
/* Parses the protobuf wire format of Example{features=1{map<string,Feature> feature=1}} (packed and unpacked
 * repeated scalars, any key order, the last duplicate key wins).  Box coordinates / classes are copied into the
 * caller's arrays (each of `capacity` elements; pass capacity 0 and NULL arrays to only count).  Errors like
 * parse_single_example: RN_EINVAL when 'image' or 'image_id' is missing, has another dtype or not exactly
 * one value, when a VarLen feature has another dtype, or on malformed wire data; RN_ENOMEM when a list is longer
 * than capacity (info is still filled with the needed lengths). */
int rn_example_parse(const uint8_t* record, size_t nbytes, rn_example_info* info, float* xmins, float* ymins,
                     float* xmaxs, float* ymaxs, int64_t* classes, int capacity);
/* Serialises the Example of TFrecordWriter._make_example (tfrecord_writer.py:27-44); boxes f32[n_boxes,4] =
 * (xmin, ymin, xmax, ymax) rows.  Feature keys are written in sorted order, repeated scalars packed.
 * Returns the size (call with out == NULL to size the buffer), or 0 when capacity is too small. */
size_t rn_example_serialize(const uint8_t* image, size_t image_bytes, int64_t image_id, const float* boxes,
                            int n_boxes, const int64_t* classes, int n_classes, uint8_t* out, size_t capacity);

/* ---------------------------------------------------------------------------------------
 * §8(f)-4  JPEG decoding for the TFRecord input path (host only; tf.io.decode_image in dataloader/tfrecord_parser.py:20-23):
 * baseline / extended-sequential Huffman JPEG, grayscale or YCbCr with 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 sampling, restart
 * intervals; libjpeg's defaults restated (islow integer IDCT, fancy chroma up-sampling, 16-bit fixed-point YCbCr->RGB).
 * Progressive / arithmetic / 12-bit / CMYK files return RN_EUNSUPPORTED with a message (corrupt ones RN_EINVAL). */
int rn_jpeg_info(const void* data, size_t len, int32_t* width, int32_t* height, int32_t* components);
int rn_jpeg_decode(const void* data, size_t len, uint8_t* rgb_out /* [height,width,3] */, size_t out_bytes);
int rn_jpeg_idct_islow(const int32_t* coef64 /* dequantized, row-major */, uint8_t* out64);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* RNET_HIP_H_ */
