"""Validation / serving preprocessing on the GPU (reference
retinanet/dataloader/preprocessing_pipeline.py:96-129 and dataloader/utils.py:58-66).

`PreprocessingPipeline(input_shape, dataloader_params).normalize_and_resize_with_pad(image)` is the
body of the exported `prepare_image` signature (export.py:244-270): image f32[h,w,3] ->
{'image': f32[H,W,3] normalised, resized with the aspect ratio kept, zero padded bottom/right,
 'resize_scale': f32[2]}.

Train side (`__call__`, reference :13-94 — SURVEY §8(f)-4's map function): normalise, random horizontal flip,
scale jitter + random crop, resize, zero pad, box transform / clip / drop of empty boxes.  The random numbers come
from a numpy Generator owned by the pipeline (TensorFlow's op-seeded streams cannot be reproduced without
TensorFlow: same distributions, different draws); the resize runs on the GPU through the same fused kernel.
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
        self.augmentation_params = params.get("augmentations") or {"use_augmentation": False}
        self.rng = np.random.default_rng(0)   # random_flip_horizontal(seed=0) / _prepare_image(seed=0)

    # ---- training sample (reference :13-94) ---------------------------------------------------------
    def _resize_to(self, image, sh, sw, th, tw):
        """Normalised image resized to [sh, sw] inside a zero [th, tw] canvas (th >= sh, tw >= sw)."""
        lib = _C.lib()
        h, w = int(image.shape[0]), int(image.shape[1])
        out = torch.empty((th, tw, 3), dtype=torch.float32, device=image.device)
        pp = self.preprocessing_params
        with torch.cuda.device(image.device):
            _C.check(lib.rn_prepare_image(_C.ptr(image), h, w, sh, sw, _C.ptr(out), th, tw, _C.f32_array(pp.mean),
                                          _C.f32_array(pp.stddev), float(pp.pixel_scale), _C.current_stream()),
                     "rn_prepare_image")
        return out

    def _prepare_image(self, image, jitter=(None, None)):
        """:13-55.  Returns (image f32[H,W,3] on the GPU, image_scale f32[2], offset i32[2], image_shape f32[2])."""
        f32 = np.float32
        target = np.asarray(self.input_shape, dtype=f32)
        image_shape = np.asarray(image.shape[:2], dtype=f32)
        scaled = target
        aug = self.augmentation_params
        if aug.get("use_augmentation"):
            jitter = (aug["scale_jitter"]["min_scale"], aug["scale_jitter"]["max_scale"])
        if jitter[0]:
            scaled = f32(self.rng.uniform(jitter[0], jitter[1])) * target
        scale = np.minimum(scaled[0] / image_shape[0], scaled[1] / image_shape[1])
        scaled = np.round(image_shape * scale)
        image_scale = (scaled / image_shape).astype(f32)
        offset = np.zeros((2,), np.int32)
        if jitter[0]:
            max_offset = np.maximum(scaled - target, f32(0))
            offset = (max_offset * self.rng.uniform(0, 1, size=2).astype(f32)).astype(np.int32)
        sh, sw = int(scaled[0]), int(scaled[1])
        th, tw = self.input_shape
        if sh <= th and sw <= tw:
            out = self._resize_to(image, sh, sw, th, tw)
        else:   # scaled past the target: resize, crop at the offset, pad what is still missing
            full = self._resize_to(image, sh, sw, max(sh, 1), max(sw, 1))
            crop = full[offset[0]:offset[0] + th, offset[1]:offset[1] + tw]
            out = torch.zeros((th, tw, 3), dtype=torch.float32, device=image.device)
            out[:crop.shape[0], :crop.shape[1]] = crop
        return out, image_scale, offset, image_shape

    def _prepare_labels(self, boxes, class_ids):
        """:57-66: clip to the canvas, corners -> (cx, cy, w, h), drop boxes without area."""
        th, tw = np.float32(self.input_shape[0]), np.float32(self.input_shape[1])
        hi = np.asarray([th, tw, th, tw], np.float32)   # tf.tile([[H, W]], [1, 2]) — as the reference writes it
        boxes = np.clip(boxes, np.float32(0), hi)
        xywh = np.concatenate([(boxes[:, :2] + boxes[:, 2:]) / np.float32(2), boxes[:, 2:] - boxes[:, :2]], axis=-1)
        keep = np.logical_and(xywh[:, 2] > 0, xywh[:, 3] > 0)
        return xywh[keep].astype(np.float32), class_ids[keep]

    def __call__(self, sample):
        """:68-94.  sample = parse_example(...) -> (image f32[H,W,3] cuda, boxes f32[n,4] (cx,cy,w,h) px, classes i32[n])."""
        image = torch.as_tensor(sample["image"], dtype=torch.float32)
        if not image.is_cuda:
            image = image.cuda()
        bbox = np.asarray(sample["objects"]["bbox"], dtype=np.float32).reshape(-1, 4)
        class_ids = np.asarray(sample["objects"]["label"]).astype(np.int32)
        aug = self.augmentation_params
        if aug.get("use_augmentation") and aug.get("horizontal_flip"):
            if self.rng.uniform() > 0.5:   # dataloader/utils.py:48-55
                image = torch.flip(image, dims=[1])
                bbox = np.stack([1 - bbox[:, 2], bbox[:, 1], 1 - bbox[:, 0], bbox[:, 3]], axis=-1).astype(np.float32)
        image, scale, offset, image_shape = self._prepare_image(image.contiguous())
        off = offset.astype(np.float32)
        bbox = np.stack([bbox[:, 0] * image_shape[1] * scale[1] - off[1],
                         bbox[:, 1] * image_shape[0] * scale[0] - off[0],
                         bbox[:, 2] * image_shape[1] * scale[1] - off[1],
                         bbox[:, 3] * image_shape[0] * scale[0] - off[0]], axis=-1).astype(np.float32)
        bbox, class_ids = self._prepare_labels(bbox, class_ids)
        return image, bbox, class_ids

    # ---- validation / serving (reference :96-129) ----------------------------------------------------
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
