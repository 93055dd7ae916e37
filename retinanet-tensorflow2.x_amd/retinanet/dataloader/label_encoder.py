"""LabelEncoder on the GPU (reference retinanet/dataloader/label_encoder.py:8-125).

The reference encodes one sample at a time inside tf.data on the host; here a whole batch of
padded ground truth is matched and encoded in two HBM-bound launches (rn_anchor_match_encode).
`encode_batch` returns the reference's target dict (`class-targets`, `box-targets` per level,
`num-positives`) as zero-copy views of the flattened outputs, plus the flattened tensors
themselves under `_flat` for the fused loss kernel.
"""
from __future__ import annotations

import math

import torch

from retinanet import _C
from retinanet.dataloader.anchor_generator import AnchorBoxGenerator


class LabelEncoder:
    def __init__(self, params, device=None, anchors=None):
        self.input_shape = list(params.input.input_shape)
        self.encoder_params = params.encoder_params
        self._min_level = params.architecture.feature_fusion.min_level
        self._max_level = params.architecture.feature_fusion.max_level
        self.anchors = anchors or AnchorBoxGenerator(*self.input_shape, self._min_level, self._max_level,
                                                     params.anchor_params, device=device)
        self._params = params
        self._device = self.anchors.boxes.device
        self._ws = None

    def encode_batch(self, gt_boxes, gt_classes, gt_counts):
        """gt_boxes f32[B,Gmax,4] (cx,cy,w,h pixels), gt_classes f32[B,Gmax], gt_counts i32[B]."""
        lib = _C.lib()
        gt_boxes = gt_boxes.to(self._device, torch.float32).contiguous()
        gt_classes = gt_classes.to(self._device, torch.float32).contiguous()
        gt_counts = gt_counts.to(self._device, torch.int32).contiguous()
        B, Gmax = gt_boxes.shape[0], gt_boxes.shape[1]
        A = self.anchors.boxes.shape[0]
        dev = self._device
        matches = torch.empty((B, A), dtype=torch.int32, device=dev)
        cls_t = torch.empty((B, A), dtype=torch.float32, device=dev)
        box_t = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        num_pos = torch.empty((B,), dtype=torch.float32, device=dev)
        need = lib.rn_match_workspace_bytes(B, Gmax)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((max(need, 256),), dtype=torch.uint8, device=dev)
        var = None
        if self.encoder_params.scale_box_targets:
            var = _C.f32_array(self.encoder_params.box_variance)
        with torch.cuda.device(dev):
            _C.check(lib.rn_anchor_match_encode(
                _C.ptr(self.anchors.boxes), A, _C.ptr(gt_boxes) if Gmax else None,
                _C.ptr(gt_classes) if Gmax else None, _C.ptr(gt_counts), B, Gmax,
                float(self.encoder_params.match_iou), float(self.encoder_params.ignore_iou), var,
                _C.ptr(matches), _C.ptr(cls_t), _C.ptr(box_t), _C.ptr(num_pos), _C.ptr(self._ws),
                self._ws.numel(), _C.current_stream()), "rn_anchor_match_encode")
        targets = {"class-targets": {}, "box-targets": {}, "num-positives": num_pos,
                   "_flat": {"matches": matches, "class-targets": cls_t, "box-targets": box_t}}
        bnd = self.anchors.anchor_boundaries
        na = self.anchors.num_anchors_per_location
        for i, level in enumerate(range(self._min_level, self._max_level + 1)):
            fh = int(math.ceil(self.input_shape[0] / 2 ** level))
            fw = int(math.ceil(self.input_shape[1] / 2 ** level))
            targets["class-targets"][str(level)] = cls_t[:, bnd[i]:bnd[i + 1]].reshape(B, fh, fw, na)
            targets["box-targets"][str(level)] = box_t[:, bnd[i]:bnd[i + 1]].reshape(B, fh, fw, 4 * na)
        return targets

    def encode_sample(self, gt_boxes, cls_ids):
        """Single-sample form of the reference's encode_sample (targets only; the image side of
        the reference's tf.data pipeline is out of scope)."""
        g = gt_boxes.reshape(1, -1, 4)
        c = cls_ids.reshape(1, -1)
        n = torch.tensor([g.shape[1]], dtype=torch.int32)
        t = self.encode_batch(g, c, n)
        out = {"class-targets": {k: v[0] for k, v in t["class-targets"].items()},
               "box-targets": {k: v[0] for k, v in t["box-targets"].items()},
               "num-positives": t["num-positives"][0], "_flat": t["_flat"]}
        return out
