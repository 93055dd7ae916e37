"""Detection post-processing on the GPU (reference
retinanet/model/layers/postprocessing_ops.py:7-561).

The reference chains four Keras layers; the same four names exist here as callables with the
same inputs/outputs so each stage can be checked on its own, and `DetectionPostProcess` is the
fused path `ModelBuilder.add_post_processing_stage` wires in: decode -> (sigmoid + threshold
compaction) -> per-(image,class) sort + NMS -> merge, without materialising the
[B,5000,K(,4)] top-k tensors.
"""
from __future__ import annotations

import torch

from retinanet import _C
from retinanet.dataloader.anchor_generator import AnchorBoxGenerator

_SUPPORTED_NMS_MODES = ["CombinedNMS", "GlobalSoftNMS", "GlobalHardNMS", "PerClassSoftNMS", "PerClassHardNMS"]


def _levels(d):
    return sorted(d.keys(), key=int)


def _level_table(preds, B, per_anchor):
    """per-level tensors [B,s,s,A*per_anchor] -> (contiguous list, anchor offsets)."""
    offs, ts = [0], []
    for lv in _levels(preds):
        t = preds[lv]
        if t.dtype != torch.float32:
            t = t.float()
        t = t.contiguous()
        ts.append(t)
        offs.append(offs[-1] + t.numel() // (B * per_anchor))
    return ts, offs


class FuseDetections:
    """postprocessing_ops.py:7-56.  Reshape+concat is pure indexing on the GPU path; this class
    materialises the fused tensors only for stage-level parity tests."""

    def __init__(self, min_level, max_level, **kwargs):
        self.min_level, self.max_level = min_level, max_level

    def __call__(self, predictions):
        cls, box = predictions["class-predictions"], predictions["box-predictions"]
        lv0 = str(self.min_level)
        B = box[lv0].shape[0]
        a = box[lv0].shape[-1] // 4
        k = cls[lv0].shape[-1] // a
        levels = [str(l) for l in range(self.min_level, self.max_level + 1)]
        return {"class_logits": torch.cat([cls[l].reshape(B, -1, k) for l in levels], dim=1),
                "encoded_boxes": torch.cat([box[l].reshape(B, -1, 4) for l in levels], dim=1)}


class TransformBoxesAndScores:
    """postprocessing_ops.py:59-117 on fused tensors: sigmoid + box decode."""

    def __init__(self, params, anchors=None, **kwargs):
        shape = params.input.input_shape
        self._h, self._w = float(shape[0]), float(shape[1])
        self._anchors = anchors or AnchorBoxGenerator(
            *shape, params.architecture.feature_fusion.min_level,
            params.architecture.feature_fusion.max_level, params.anchor_params)
        self._var = (list(params.encoder_params.box_variance)
                     if params.encoder_params.scale_box_targets else None)

    def decode(self, box_levels, offs, B):
        lib = _C.lib()
        A = offs[-1]
        boxes = torch.empty((B, A, 4), dtype=torch.float32, device=box_levels[0].device)
        with torch.cuda.device(boxes.device):
            _C.check(lib.rn_decode_boxes(_C.ptr_array(box_levels), _C.i64_array(offs), len(box_levels), B,
                                         _C.ptr(self._anchors.boxes), _C.f32_array(self._var), self._h, self._w,
                                         _C.ptr(boxes), _C.current_stream()), "rn_decode_boxes")
        return boxes

    def __call__(self, predictions):
        logits = predictions["class_logits"].float().contiguous()
        enc = predictions["encoded_boxes"].float().contiguous()
        B, A, K = logits.shape
        lib = _C.lib()
        scores = torch.empty_like(logits)
        with torch.cuda.device(logits.device):
            _C.check(lib.rn_sigmoid_scores(_C.ptr_array([logits]), _C.i64_array([0, A]), 1, B, K,
                                           _C.ptr(scores), _C.current_stream()), "rn_sigmoid_scores")
        return {"scores": scores, "boxes": self.decode([enc], [0, A], B)}


class FilterTopKDetections:
    """postprocessing_ops.py:120-173 (per-class filter).  Output order is canonical:
    descending score, ties by ascending anchor index."""

    def __init__(self, top_k=100, filter_per_class=True, **kwargs):
        if not filter_per_class:
            raise NotImplementedError("filter_per_class=false is unused by every shipped config")
        self.top_k = top_k
        self._ws = None

    def __call__(self, predictions):
        lib = _C.lib()
        scores = predictions["scores"].float().contiguous()
        boxes = predictions["boxes"].float().contiguous()
        B, A, K = scores.shape
        k = min(self.top_k, A)
        out_s = torch.empty((B, k, K), dtype=torch.float32, device=scores.device)
        out_i = torch.empty((B, k, K), dtype=torch.int32, device=scores.device)
        need = lib.rn_topk_workspace_bytes(B, A, K)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=scores.device)
        with torch.cuda.device(scores.device):
            _C.check(lib.rn_topk_per_class(_C.ptr(scores), B, A, K, k, _C.ptr(out_s), _C.ptr(out_i),
                                           _C.ptr(self._ws), self._ws.numel(), _C.current_stream()),
                     "rn_topk_per_class")
        idx = out_i.long()
        gathered = torch.stack([boxes[b][idx[b]] for b in range(B)], dim=0)  # [B,k,K,4]
        return {"scores": out_s, "boxes": gathered, "indices": out_i}


class GenerateDetections:
    """postprocessing_ops.py:176-561, per-class modes (the ones shipped configs select)."""
    _SUPPORTED_NMS_MODES = _SUPPORTED_NMS_MODES

    def __init__(self, iou_threshold=0.5, score_threshold=0.05, max_detections=100, soft_nms_sigma=None,
                 num_classes=None, mode="CombinedNMS", **kwargs):
        if mode not in _SUPPORTED_NMS_MODES:
            raise AssertionError("Requested unsupported mode: {}, available modes are: {}".format(
                mode, _SUPPORTED_NMS_MODES))
        if mode not in ("PerClassHardNMS", "PerClassSoftNMS"):
            raise NotImplementedError(f"NMS mode {mode} is not selected by any shipped config (SURVEY a15)")
        self.iou_threshold = float(iou_threshold)
        self.score_threshold = float(score_threshold)
        self.max_detections = int(max_detections)
        self.soft_nms_sigma = float(soft_nms_sigma or 0.0)
        self.num_classes = num_classes
        self.mode = mode
        self._ws = None

    @property
    def sigma(self):
        return self.soft_nms_sigma if self.mode == "PerClassSoftNMS" else 0.0

    def __call__(self, predictions):
        lib = _C.lib()
        scores = predictions["scores"].float().contiguous()
        boxes = predictions["boxes"].float().contiguous()
        B, n, K = scores.shape
        if boxes.dim() == 3:
            boxes = boxes[:, :, None, :].expand(B, n, K, 4).contiguous()
        dev = scores.device
        md = self.max_detections
        out = _alloc_detections(B, md, dev)
        need = lib.rn_nms_workspace_bytes(B, n, K, md)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _C.check(lib.rn_nms_per_class(_C.ptr(scores), _C.ptr(boxes), B, n, K, self.iou_threshold,
                                          self.score_threshold, self.sigma, md, _C.ptr(out["boxes"]),
                                          _C.ptr(out["scores"]), _C.ptr(out["classes"]),
                                          _C.ptr(out["valid_detections"]), _C.ptr(self._ws), self._ws.numel(),
                                          _C.current_stream()), "rn_nms_per_class")
        return out


def _alloc_detections(B, md, dev):
    return {"boxes": torch.empty((B, md, 4), dtype=torch.float32, device=dev),
            "scores": torch.empty((B, md), dtype=torch.float32, device=dev),
            "classes": torch.empty((B, md), dtype=torch.int32, device=dev),
            "valid_detections": torch.empty((B,), dtype=torch.int32, device=dev)}


class DetectionPostProcess:
    """The fused stage behind ModelBuilder.add_post_processing_stage (model/builder.py:153-190)."""

    def __init__(self, params, anchors=None):
        inf = params.inference
        self._tb = TransformBoxesAndScores(params, anchors=anchors)
        self._gen = GenerateDetections(iou_threshold=inf.iou_threshold, score_threshold=inf.score_threshold,
                                       max_detections=inf.max_detections, soft_nms_sigma=inf.soft_nms_sigma,
                                       num_classes=params.architecture.head.num_classes, mode=inf.mode)
        if not inf.filter_per_class and inf.pre_nms_top_k > 0:
            raise NotImplementedError("inference.filter_per_class=false is unused by every shipped config")
        self._top_k = int(inf.pre_nms_top_k)
        self._K = int(params.architecture.head.num_classes)
        self._ws = None
        self._out = None
        self._boxes = None

    def __call__(self, predictions):
        lib = _C.lib()
        lv0 = _levels(predictions["box-predictions"])[0]
        B = predictions["box-predictions"][lv0].shape[0]
        box_levels, offs = _level_table(predictions["box-predictions"], B, 4)
        cls_levels, offs_c = _level_table(predictions["class-predictions"], B, self._K)
        if offs != offs_c:
            raise ValueError("class and box predictions disagree on anchors per level")
        A = offs[-1]
        dev = box_levels[0].device
        boxes = self._tb.decode(box_levels, offs, B)
        g = self._gen
        md = g.max_detections
        if self._out is None or self._out["scores"].shape[0] != B:
            self._out = _alloc_detections(B, md, dev)
        out = self._out
        need = lib.rn_detect_workspace_bytes(B, A, self._K, md)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _C.check(lib.rn_detect_per_class(
                _C.ptr_array(cls_levels), _C.i64_array(offs), len(cls_levels), B, self._K, _C.ptr(boxes),
                self._top_k, g.iou_threshold, g.score_threshold, g.sigma, md, _C.ptr(out["boxes"]),
                _C.ptr(out["scores"]), _C.ptr(out["classes"]), _C.ptr(out["valid_detections"]),
                _C.ptr(self._ws), self._ws.numel(), _C.current_stream()), "rn_detect_per_class")
        return out
