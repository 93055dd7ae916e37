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

    def decode(self, box_levels, offs, B, out=None):
        """out: a [B, A, 4] f32 buffer to decode into (the serving stage keeps one: no allocation per call, stable
        addresses for a captured HIP graph)"""
        lib = _C.lib()
        A = offs[-1]
        boxes = out if out is not None else torch.empty((B, A, 4), dtype=torch.float32, device=box_levels[0].device)
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
        self.top_k = top_k
        self.filter_per_class = filter_per_class
        self._ws = None

    def _filter_global(self, scores, boxes):
        """postprocessing_ops.py:149-161: top-k over the flattened (anchor, class) scores; an anchor
        appears once per class that made the cut (the reference gathers rows by `index // K`)."""
        lib = _C.lib()
        B, A, K = scores.shape
        k = min(self.top_k, A * K)
        flat = scores.reshape(B, A * K, 1)
        out_s = torch.empty((B, k, 1), dtype=torch.float32, device=scores.device)
        out_i = torch.empty((B, k, 1), dtype=torch.int32, device=scores.device)
        need = lib.rn_topk_workspace_bytes(B, A * K, 1)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=scores.device)
        with torch.cuda.device(scores.device):
            _C.check(lib.rn_topk_per_class(_C.ptr(flat), B, A * K, 1, k, _C.ptr(out_s), _C.ptr(out_i),
                                           _C.ptr(self._ws), self._ws.numel(), _C.current_stream()),
                     "rn_topk_per_class")
        anchor = (out_i[:, :, 0].long() // K)
        bi = torch.arange(B, device=scores.device)[:, None]
        return {"scores": scores[bi, anchor].contiguous(), "boxes": boxes[bi, anchor].contiguous(),
                "indices": out_i[:, :, 0]}

    def __call__(self, predictions):
        lib = _C.lib()
        scores = predictions["scores"].float().contiguous()
        boxes = predictions["boxes"].float().contiguous()
        if not self.filter_per_class:
            return self._filter_global(scores, boxes)
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
    """postprocessing_ops.py:176-561.  PerClassHardNMS / PerClassSoftNMS (the modes every shipped
    config selects), GlobalHardNMS / GlobalSoftNMS (:244-286) and CombinedNMS (:219-242).

    `strict_reference=True` reproduces the reference's GlobalHardNMS quirk off-TPU: it passes
    iou_threshold = 1.0 when sigma is 0 (:253, inverted with respect to :448), so nothing is ever
    suppressed; the default applies the configured threshold."""
    _SUPPORTED_NMS_MODES = _SUPPORTED_NMS_MODES

    def __init__(self, iou_threshold=0.5, score_threshold=0.05, max_detections=100, soft_nms_sigma=None,
                 num_classes=None, mode="CombinedNMS", strict_reference=False, **kwargs):
        if mode not in _SUPPORTED_NMS_MODES:
            raise AssertionError("Requested unsupported mode: {}, available modes are: {}".format(
                mode, _SUPPORTED_NMS_MODES))
        self.iou_threshold = float(iou_threshold)
        self.score_threshold = float(score_threshold)
        self.max_detections = int(max_detections)
        self.soft_nms_sigma = float(soft_nms_sigma or 0.0)
        self.num_classes = num_classes
        self.mode = mode
        self.strict_reference = bool(strict_reference)
        self._ws = None
        self._ws2 = None

    @property
    def sigma(self):
        return self.soft_nms_sigma if self.mode in ("PerClassSoftNMS", "GlobalSoftNMS") else 0.0

    def _run(self, scores, boxes, iou_thr, sigma, want_index=False):
        lib = _C.lib()
        B, n, K = scores.shape
        dev = scores.device
        md = self.max_detections
        out = _alloc_detections(B, md, dev)
        idx = torch.empty((B, md), dtype=torch.int32, device=dev) if want_index else None
        need = lib.rn_nms_workspace_bytes(B, n, K, md)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _C.check(lib.rn_nms_per_class(_C.ptr(scores), _C.ptr(boxes), B, n, K, iou_thr, self.score_threshold,
                                          sigma, md, _C.ptr(out["boxes"]), _C.ptr(out["scores"]),
                                          _C.ptr(out["classes"]), _C.ptr(idx), _C.ptr(out["valid_detections"]),
                                          _C.ptr(self._ws), self._ws.numel(), _C.current_stream()),
                     "rn_nms_per_class")
        return out, idx

    def _global(self, scores, boxes):
        lib = _C.lib()
        if boxes.dim() != 3:
            raise ValueError("Global NMS modes take class-agnostic boxes [B,n,4] "
                             "(inference.filter_per_class must be false when pre_nms_top_k > 0)")
        B, n, K = scores.shape
        mx = torch.empty((B, n, 1), dtype=torch.float32, device=scores.device)
        arg = torch.empty((B, n), dtype=torch.int32, device=scores.device)
        with torch.cuda.device(scores.device):
            _C.check(lib.rn_rowmax_argmax(_C.ptr(scores), B * n, K, _C.ptr(mx), _C.ptr(arg), _C.current_stream()),
                     "rn_rowmax_argmax")
        sigma = self.sigma
        iou = self.iou_threshold
        if not sigma and self.strict_reference:
            iou = 1.0   # postprocessing_ops.py:253
        out, idx = self._run(mx, boxes.reshape(B, n, 1, 4).contiguous(), iou, sigma, want_index=True)
        cls = torch.gather(arg, 1, idx.clamp(min=0).long()).to(torch.int32)
        out["classes"] = torch.where(idx >= 0, cls, torch.full_like(cls, -1))
        return out

    def __call__(self, predictions):
        scores = predictions["scores"].float().contiguous()
        boxes = predictions["boxes"].float().contiguous()
        if self.mode in ("GlobalHardNMS", "GlobalSoftNMS"):
            return self._global(scores, boxes)
        B, n, K = scores.shape
        if boxes.dim() == 3:
            boxes = boxes[:, :, None, :].expand(B, n, K, 4).contiguous()
        if self.mode == "CombinedNMS":
            # tf.image.combined_non_max_suppression: per-class hard NMS (<= max_detections per class),
            # best max_detections overall, boxes clipped, ZERO padding and float classes
            out, _ = self._run(scores, boxes, self.iou_threshold, 0.0)
            pad = out["scores"] < 0
            out["scores"] = torch.where(pad, torch.zeros_like(out["scores"]), out["scores"])
            out["classes"] = torch.where(pad, torch.zeros_like(out["classes"]), out["classes"]).float()
            return out
        out, _ = self._run(scores, boxes, self.iou_threshold, self.sigma)
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
        self._fused = inf.mode in ("PerClassHardNMS", "PerClassSoftNMS") and (inf.filter_per_class or
                                                                                 inf.pre_nms_top_k <= 0)
        self._filter = (FilterTopKDetections(top_k=inf.pre_nms_top_k, filter_per_class=inf.filter_per_class)
                        if inf.pre_nms_top_k > 0 else None)
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
        if self._boxes is None or tuple(self._boxes.shape) != (B, A, 4) or self._boxes.device != dev:
            self._boxes = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        boxes = self._tb.decode(box_levels, offs, B, out=self._boxes)
        if not self._fused:
            # stage-by-stage path for the modes no shipped config selects (a15)
            lib0 = _C.lib()
            scores = torch.empty((B, A, self._K), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _C.check(lib0.rn_sigmoid_scores(_C.ptr_array(cls_levels), _C.i64_array(offs), len(cls_levels), B,
                                                self._K, _C.ptr(scores), _C.current_stream()), "rn_sigmoid_scores")
            x = {"scores": scores, "boxes": boxes}
            if self._filter is not None:
                x = self._filter(x)
            return self._gen(x)
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
