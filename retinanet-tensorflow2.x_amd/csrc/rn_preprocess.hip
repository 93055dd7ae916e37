// rn_preprocess.hip — §8(f)-1: the `prepare_image` serving signature / validation preprocessing
// (retinanet/dataloader/preprocessing_pipeline.py:96-121 normalize_and_resize_with_pad,
// retinanet/dataloader/utils.py:58-66 normalize_image): (x / pixel_scale - mean) / stddev, aspect
// preserving bilinear resize (tf.image.resize, TF2 half-pixel centres, no antialias), zero pad at the
// bottom/right to the network input size.  One fused pass: each output pixel normalises its four
// source taps and interpolates in fp32 in TensorFlow's op order (top row lerp, bottom row lerp,
// vertical lerp); -ffp-contract=off keeps it bit-identical to the numpy oracle.  HBM-bound:
// 12 B written per output pixel, <= 48 B read.
#include "rn_common.h"

struct PrepArgs {
  const float* img;
  float* out;
  int h, w, sh, sw, th, tw;
  float scale_y, scale_x;  // in / out (float32, as TF computes it)
  float pixel_scale, mean[3], stddev[3];
};

__global__ void __launch_bounds__(256) prepare_image_kernel(PrepArgs a) {
  const int total = a.th * a.tw;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int y = i / a.tw, x = i - y * a.tw;
    float o[3] = {0.0f, 0.0f, 0.0f};
    if (y < a.sh && x < a.sw) {
      const float in_y = ((float)y + 0.5f) * a.scale_y - 0.5f;
      const float in_x = ((float)x + 0.5f) * a.scale_x - 0.5f;
      const float fy = floorf(in_y), fx = floorf(in_x);
      const int y0 = max((int)fy, 0), y1 = min((int)ceilf(in_y), a.h - 1);
      const int x0 = max((int)fx, 0), x1 = min((int)ceilf(in_x), a.w - 1);
      const float ly = in_y - fy, lx = in_x - fx;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float tl = (a.img[((long long)y0 * a.w + x0) * 3 + c] / a.pixel_scale - a.mean[c]) / a.stddev[c];
        const float tr = (a.img[((long long)y0 * a.w + x1) * 3 + c] / a.pixel_scale - a.mean[c]) / a.stddev[c];
        const float bl = (a.img[((long long)y1 * a.w + x0) * 3 + c] / a.pixel_scale - a.mean[c]) / a.stddev[c];
        const float br = (a.img[((long long)y1 * a.w + x1) * 3 + c] / a.pixel_scale - a.mean[c]) / a.stddev[c];
        const float top = tl + (tr - tl) * lx;
        const float bot = bl + (br - bl) * lx;
        o[c] = top + (bot - top) * ly;
      }
    }
    a.out[(long long)i * 3 + 0] = o[0];
    a.out[(long long)i * 3 + 1] = o[1];
    a.out[(long long)i * 3 + 2] = o[2];
  }
}

extern "C" int rn_prepare_image(const float* image, int h, int w, int scaled_h, int scaled_w, float* out,
                                int target_h, int target_w, const float* mean, const float* stddev,
                                float pixel_scale, void* stream) {
  RN_CHECK_ARG(image && out && mean && stddev && h > 0 && w > 0 && scaled_h > 0 && scaled_w > 0 &&
                   scaled_h <= target_h && scaled_w <= target_w,
               "rn_prepare_image: bad argument (the scaled image must fit the target)");
  PrepArgs a;
  a.img = image; a.out = out;
  a.h = h; a.w = w; a.sh = scaled_h; a.sw = scaled_w; a.th = target_h; a.tw = target_w;
  a.scale_y = (float)h / (float)scaled_h;
  a.scale_x = (float)w / (float)scaled_w;
  a.pixel_scale = pixel_scale;
  for (int c = 0; c < 3; ++c) {
    a.mean[c] = mean[c];
    a.stddev[c] = stddev[c];
  }
  const int total = target_h * target_w;
  hipLaunchKernelGGL(prepare_image_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  RN_CHECK_LAUNCH();
  return RN_OK;
}


// ---- (f)-2  COCOEvaluator.accumulate_results (retinanet/eval/coco_evaluator.py:95-134) ----------------
// boxes /= tile(resize_scale / input_shape, 2); int32 truncation; (x1,y1,x2,y2) -> (x,y,w,h) on the truncated
// values; class id through the sorted-name lookup table.  One thread per detection slot.
__global__ void __launch_bounds__(256)
coco_accumulate_kernel(const float4* __restrict__ boxes, const int* __restrict__ classes, const int* __restrict__ valid,
                       const float2* __restrict__ resize_scale, float in_h, float in_w, const int* __restrict__ lut,
                       int num_classes, int B, int D, int rescale, int4* __restrict__ out_bbox,
                       int* __restrict__ out_cat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  if (d >= valid[b]) {
    out_bbox[i] = make_int4(0, 0, 0, 0);
    out_cat[i] = -1;
    return;
  }
  float4 v = boxes[i];
  if (rescale) {
    const float2 rs = resize_scale[b];
    const float s0 = rs.x / in_h, s1 = rs.y / in_w;   // resize_scales[i] / self._input_shape  (:118)
    v.x = v.x / s0; v.y = v.y / s1; v.z = v.z / s0; v.w = v.w / s1;   // tiled x2 (:119-122)
  }
  const int x1 = (int)v.x, y1 = (int)v.y, x2 = (int)v.z, y2 = (int)v.w;   // np.int32(): truncation (:124)
  out_bbox[i] = make_int4(x1, y1, x2 - x1, y2 - y1);                          // :125
  const int c = classes[i];
  out_cat[i] = (lut && c >= 0 && c < num_classes) ? lut[c] : c;
}

extern "C" int rn_coco_accumulate(const float* boxes, const int32_t* classes, const int32_t* valid,
                                  const float* resize_scale, float input_h, float input_w, const int32_t* class_lut,
                                  int num_classes, int B, int D, int rescale, int32_t* out_bbox, int32_t* out_category,
                                  void* stream) {
  RN_CHECK_ARG(boxes && classes && valid && out_bbox && out_category && B > 0 && D > 0 && (!rescale || resize_scale) &&
                   input_h > 0 && input_w > 0, "rn_coco_accumulate: bad argument");
  hipLaunchKernelGGL(coco_accumulate_kernel, dim3((B * D + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)boxes, classes, valid, (const float2*)resize_scale, input_h, input_w, class_lut,
                     num_classes, B, D, rescale, (int4*)out_bbox, out_category);
  RN_CHECK_LAUNCH();
  return RN_OK;
}
