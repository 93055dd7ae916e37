"""Box helpers and `normalize_image` of the reference (retinanet/dataloader/utils.py:4-66) on torch tensors.

`normalize_image` is the exported one (dataloader/__init__.py): `(image / pixel_scale - mean) / stddev`.  On a GPU
tensor it runs the fused prepare-image kernel at scale 1 (normalise + identity resize: with equal input and output
sizes the half-pixel bilinear weights are exactly (1, 0), so the kernel's normalisation arithmetic — the same
float32 divide / subtract / divide as the reference — is all that happens)."""
from __future__ import annotations

import torch


def swap_xy(boxes):
    return torch.stack([boxes[:, 1], boxes[:, 0], boxes[:, 3], boxes[:, 2]], dim=-1)


def convert_to_xywh(boxes):
    return torch.cat([(boxes[..., :2] + boxes[..., 2:]) / 2.0, boxes[..., 2:] - boxes[..., :2]], dim=-1)


def convert_to_corners(boxes):
    return torch.cat([boxes[..., :2] - boxes[..., 2:] / 2.0, boxes[..., :2] + boxes[..., 2:] / 2.0], dim=-1)


def normalize_image(image, mean, stddev, pixel_scale):
    from retinanet import _C
    image = torch.as_tensor(image, dtype=torch.float32)
    if image.dim() != 3 or image.shape[2] != 3:
        raise ValueError("image must be [h, w, 3]")
    if not image.is_cuda:
        image = image.cuda()
    image = image.contiguous()
    h, w = int(image.shape[0]), int(image.shape[1])
    out = torch.empty_like(image)
    with torch.cuda.device(image.device):
        _C.check(_C.lib().rn_prepare_image(_C.ptr(image), h, w, h, w, _C.ptr(out), h, w, _C.f32_array(mean),
                                           _C.f32_array(stddev), float(pixel_scale), _C.current_stream()),
                 "rn_prepare_image")
    return out
