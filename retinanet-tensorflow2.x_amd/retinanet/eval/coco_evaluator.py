"""COCOEvaluator — the detection accumulation step of the reference's evaluator
(retinanet/eval/coco_evaluator.py:23-164) behind the same class surface.

`accumulate_results` (:95-134) is served by `rn_coco_accumulate` on the detections as they come
off the NMS kernels (rescale to original-image pixels, int32 truncation, xyxy -> xywh, class id
remap); the per-detection dict list and the JSON dump stay host code.  The category table comes
from the annotation JSON itself (`categories`).  `evaluate()` (:136-157) runs pycocotools' `COCOeval` when
pycocotools is installed and otherwise the NumPy restatement of it in `retinanet/eval/cocoeval.py` (bbox only,
same twelve statistics).
"""
from __future__ import annotations

import json
import os

import torch

from retinanet import _C


class COCOEvaluator:
    def __init__(self, input_shape, annotation_file_path=None, prediction_file_path="predictions.json",
                 remap_class_ids=False, categories=None):
        self._input_shape = [float(input_shape[0]), float(input_shape[1])]
        self.annotation_file_path = annotation_file_path
        self.prediction_file_path = os.path.normpath(prediction_file_path)
        self._remap_class_ids = bool(remap_class_ids)
        if categories is None:
            if annotation_file_path is None:
                raise ValueError("either `annotation_file_path` or `categories` is required")
            with open(annotation_file_path) as fp:
                categories = json.load(fp)["categories"]
        # coco_evaluator.py:38-52: contiguous ids follow the alphabetical order of the category names
        sorted_classes = sorted(c["name"] for c in categories)
        self._class_name_to_orig_class_id = {c["name"]: c["id"] for c in categories}
        self._sorted_class_name_to_class_id = {n: i for i, n in enumerate(sorted_classes)}
        self._sorted_class_id_to_class_name = {i: n for i, n in enumerate(sorted_classes)}
        self._lut = [self._class_name_to_orig_class_id[n] for n in sorted_classes]
        self._lut_dev = {}
        self._processed_detections = []

    def _maybe_remap_class_ids(self, class_id):
        if self._remap_class_ids:
            return self._class_name_to_orig_class_id[self._sorted_class_id_to_class_name[class_id]]
        return class_id

    def accumulate_results(self, results, rescale_detections=True):
        """results: {'image_id': [B], 'detections': {boxes, scores, classes, valid_detections} (device
        tensors), 'resize_scale': f32[B,2]} — coco_evaluator.py:95-134."""
        det = results["detections"]
        boxes = det["boxes"].to(torch.float32).contiguous()
        dev = boxes.device
        B, D = boxes.shape[0], boxes.shape[1]
        classes = det["classes"].to(device=dev, dtype=torch.int32).contiguous()
        valid = det["valid_detections"].to(device=dev, dtype=torch.int32).contiguous()
        scale = torch.as_tensor(results["resize_scale"], dtype=torch.float32, device=dev).reshape(B, 2).contiguous()
        lut = None
        if self._remap_class_ids:
            if dev not in self._lut_dev:
                self._lut_dev[dev] = torch.tensor(self._lut, dtype=torch.int32, device=dev)
            lut = self._lut_dev[dev]
        bbox = torch.empty((B, D, 4), dtype=torch.int32, device=dev)
        cat = torch.empty((B, D), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _C.check(_C.lib().rn_coco_accumulate(_C.ptr(boxes), _C.ptr(classes), _C.ptr(valid), _C.ptr(scale),
                                                 self._input_shape[0], self._input_shape[1],
                                                 _C.ptr(lut) if lut is not None else None, len(self._lut), B, D,
                                                 1 if rescale_detections else 0, _C.ptr(bbox), _C.ptr(cat),
                                                 _C.current_stream()), "rn_coco_accumulate")
        bbox_h, cat_h = bbox.cpu().tolist(), cat.cpu().tolist()
        scores_h, valid_h = det["scores"].to(torch.float32).cpu().tolist(), valid.cpu().tolist()
        image_ids = [int(x) for x in (results["image_id"].tolist() if hasattr(results["image_id"], "tolist")
                                      else results["image_id"])]
        for i in range(B):
            for d in range(valid_h[i]):
                self._processed_detections.append({"image_id": image_ids[i], "category_id": cat_h[i][d],
                                                   "bbox": bbox_h[i][d], "score": float(scores_h[i][d])})

    def evaluate(self, print_fn=None):
        with open(self.prediction_file_path, "w") as f:
            json.dump(self._processed_detections, f, indent=4)
        try:
            from pycocotools.coco import COCO
            from pycocotools.cocoeval import COCOeval
            gt = COCO(self.annotation_file_path)
            ev = COCOeval(gt, gt.loadRes(self.prediction_file_path), "bbox")
            ev.evaluate()
            ev.accumulate()
            ev.summarize()
        except ImportError:
            from retinanet.eval.cocoeval import COCO, COCOeval
            gt = COCO(self.annotation_file_path)
            ev = COCOeval(gt, gt.loadRes(self.prediction_file_path) if self._processed_detections
                          else COCO(dataset={"images": list(gt.imgs.values()), "categories": list(gt.cats.values()),
                                             "annotations": []}), "bbox")
            ev.evaluate()
            ev.accumulate()
            ev.summarize(print_fn=print_fn)
        return {"AP-IoU=0.50:0.95": ev.stats[0], "AP-IoU=0.50": ev.stats[1], "AP-IoU=0.75": ev.stats[2],
                "AR-(all)-IoU=0.50:0.95": ev.stats[6], "AR-(L)-IoU=0.50:0.95": ev.stats[-1]}

    @property
    def processed_detections(self):
        return self._processed_detections
