"""Validation / serving preprocessing on the GPU (reference
retinanet/dataloader/preprocessing_pipeline.py:96-129 and dataloader/utils.py:58-66).

`PreprocessingPipeline(input_shape, dataloader_params).normalize_and_resize_with_pad(image)` is the
body of the exported `prepare_image` signature (export.py:244-270): image f32[h,w,3] ->
{'image': f32[H,W,3] normalised, resized with the aspect ratio kept, zero padded bottom/right,
 'resize_scale': f32[2]}.  The train-side augmentation (flip, scale jitter, crop) stays a host /
tf.data concern and is out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from retinanet import _C


def scaled_shape_and_scale(h, w, target_h, target_w):
    """preprocessing_pipeline.py:98-103 in float32: round(shape * min(target/shape)), scale = scaled/shape."""
    shape = np.asarray([h, w], dtype=np.float32)
    ratio = np.minimum(np.float32(target_h) / shape[0], np.float32(target_w) / shape[1])
    scaled = np.round(shape * ratio)          # tf.round: half to even, like np.round
    return int(scaled[0]), int(scaled[1]), (scaled / shape).astype(np.float32)


class PreprocessingPipeline:
    def __init__(self, input_shape, params):
        self.input_shape = list(input_shape)
        self.preprocessing_params = params.preprocessing

    def normalize_and_resize_with_pad(self, image):
        lib = _C.lib()
        if image.dim() != 3 or image.shape[2] != 3:
            raise ValueError("image must be [h, w, 3]")
        image = image.to(torch.float32).contiguous()
        if not image.is_cuda:
            image = image.cuda()
        h, w = int(image.shape[0]), int(image.shape[1])
        th, tw = self.input_shape
        sh, sw, scale = scaled_shape_and_scale(h, w, th, tw)
        out = torch.empty((th, tw, 3), dtype=torch.float32, device=image.device)
        pp = self.preprocessing_params
        with torch.cuda.device(image.device):
            _C.check(lib.rn_prepare_image(_C.ptr(image), h, w, sh, sw, _C.ptr(out), th, tw, _C.f32_array(pp.mean),
                                          _C.f32_array(pp.stddev), float(pp.pixel_scale), _C.current_stream()),
                     "rn_prepare_image")
        return {"image": out, "resize_scale": torch.from_numpy(scale)}

    def preprocess_val_sample(self, sample):
        p = self.normalize_and_resize_with_pad(sample["image"])
        return {"image": p["image"], "image_id": sample["image_id"], "resize_scale": p["resize_scale"]}
