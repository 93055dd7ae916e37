"""Dump what the soft-NMS stage of the config-4 serving bench sees for image 0: decoded boxes [A,4] and the sigmoid scores of
a few classes [A,n] (float32, .npz) — input of tools/probes/soft_nms_sim.py --dump.
python tools/probes/c4_dump_lists.py --out gpurun_out/nms/lists.npz"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from retinanet.model.layers.postprocessing_ops import DetectionPostProcess, _level_table

ap = argparse.ArgumentParser()
ap.add_argument("--infer-batch", type=int, default=8)
ap.add_argument("--logit-std", type=float, default=1.0)
ap.add_argument("--classes", type=int, default=16)
ap.add_argument("--out", default="gpurun_out/nms/lists.npz")
a = ap.parse_args()
dev = torch.device("cuda:0")
p4, bi, mi, x = bench.build_config4_serving(a, dev)
preds = mi(x)
B = x.shape[0]
post = DetectionPostProcess(p4)
box_levels, offs = _level_table(preds["box-predictions"], B, 4)
boxes = post._tb.decode(box_levels, offs, B, out=torch.empty((B, offs[-1], 4), dtype=torch.float32, device=dev))
K = p4.architecture.head.num_classes
cls_levels, _ = _level_table(preds["class-predictions"], B, K)
logits = torch.cat([t[0].reshape(-1, K) for t in cls_levels], 0)
scores = torch.sigmoid(logits[:, :a.classes].float())
os.makedirs(os.path.dirname(a.out), exist_ok=True)
np.savez_compressed(a.out, boxes=boxes[0].clamp(0, 1).cpu().numpy(), scores=scores.cpu().numpy(),
                    score_threshold=p4.inference.score_threshold, sigma=p4.inference.soft_nms_sigma,
                    top_k=p4.inference.pre_nms_top_k, max_det=p4.inference.max_detections)
print("wrote", a.out, tuple(boxes[0].shape), tuple(scores.shape), "candidates per class",
      (scores > p4.inference.score_threshold).sum(0).tolist())
