"""Bounding-box COCO evaluation in NumPy — what `pycocotools.cocoeval.COCOeval(gt, dt, 'bbox')` computes for the
reference's evaluator (retinanet/eval/coco_evaluator.py:136-157; requirements.txt:4 pins pycocotools==2.0.2, which is
not installed here and cannot be: no network).  SURVEY §8(f)-2.

The algorithm is restated from pycocotools' published definition (cocoeval.py `evaluate` / `computeIoU` /
`evaluateImg` / `accumulate` / `summarize`, bbox branch):
  * parameters: IoU thresholds 0.50:0.05:0.95, 101 recall thresholds 0:0.01:1, maxDets (1, 10, 100), area ranges
    all / small (< 32^2) / medium / large (> 96^2), per-category evaluation;
  * per (image, category): detections sorted by score (stable), at most maxDets[-1]; IoU of [x, y, w, h] boxes, against a
    crowd ground truth IoU = intersection / detection area; ground truths outside the area range or crowd are
    "ignore" and are sorted last; greedy matching per IoU threshold in score order — a detection takes the unmatched
    (or crowd) ground truth of highest IoU >= threshold, preferring non-ignored ones, the LAST one among equal IoUs;
    unmatched detections outside the area range are ignored;
  * accumulate: per (category, area, maxDet) detections of all images merged by score (stable), cumulative TP / FP,
    precision made monotonically non-increasing from the right, sampled at the recall thresholds with
    searchsorted(side='left'); recall = final TP / number of non-ignored ground truths; -1 where there is no ground truth;
  * summarize: the twelve numbers, each the mean over the entries > -1.
`COCO` below is the part of `pycocotools.coco.COCO` the evaluation touches (the annotation index and `loadRes` for
bbox results: area = w*h, id = 1..n, iscrowd = 0).  PARITY UNPINNED against pycocotools itself (absent); pinned by
hand-computed cases and invariances in tests/test_cocoeval_cpu.py."""
from __future__ import annotations

import json
from collections import defaultdict

import numpy as np


class COCO:
    def __init__(self, annotation_file=None, dataset=None):
        if dataset is None:
            with open(annotation_file) as f:
                dataset = json.load(f)
        self.dataset = dataset
        self.imgs = {im["id"]: im for im in dataset.get("images", [])}
        self.cats = {c["id"]: c for c in dataset.get("categories", [])}
        self.anns = {}
        self.img_cat_to_anns = defaultdict(list)
        for a in dataset.get("annotations", []):
            self.anns[a["id"]] = a
            self.img_cat_to_anns[(a["image_id"], a["category_id"])].append(a)

    def getImgIds(self):
        return list(self.imgs)

    def getCatIds(self):
        return list(self.cats)

    def loadRes(self, results):
        """results: path of the prediction JSON or the list itself ([{image_id, category_id, bbox, score}, ...])"""
        if isinstance(results, str):
            with open(results) as f:
                results = json.load(f)
        if not isinstance(results, list):
            raise AssertionError("results in not an array of objects")
        unknown = set(r["image_id"] for r in results) - set(self.imgs)
        if unknown:
            raise AssertionError("Results do not correspond to current coco set")
        anns = []
        for i, r in enumerate(results):
            a = dict(r)
            x, y, w, h = a["bbox"]
            a["area"], a["id"], a["iscrowd"] = w * h, i + 1, 0
            anns.append(a)
        return COCO(dataset={"images": list(self.imgs.values()), "categories": list(self.cats.values()),
                             "annotations": anns})


def bbox_iou(dt, gt, iscrowd):
    """dt [D,4], gt [G,4] as (x, y, w, h); crowd columns: intersection / detection area (maskUtils.iou)"""
    dt, gt = np.asarray(dt, np.float64).reshape(-1, 4), np.asarray(gt, np.float64).reshape(-1, 4)
    if dt.shape[0] == 0 or gt.shape[0] == 0:
        return np.zeros((dt.shape[0], gt.shape[0]))
    iw = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    ih = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
    da, ga = (dt[:, 2] * dt[:, 3])[:, None], (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(np.asarray(iscrowd, bool)[None, :], da, da + ga - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union > 0, inter / union, 0.0)


class Params:
    def __init__(self):
        self.imgIds, self.catIds = [], []
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ["all", "small", "medium", "large"]
        self.useCats = 1


class COCOeval:
    def __init__(self, cocoGt, cocoDt, iouType="bbox"):
        if iouType != "bbox":
            raise ValueError("only iouType='bbox' is evaluated by the reference (coco_evaluator.py:147)")
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.params = Params()
        self.params.imgIds = sorted(cocoGt.getImgIds())
        self.params.catIds = sorted(cocoGt.getCatIds())
        self.evalImgs, self.eval, self.stats = [], {}, []

    # ---- per image ----------------------------------------------------------------------------------------------
    def _evaluate_img(self, img_id, cat_id, ious, gts, dts, a_rng, max_det):
        if not gts and not dts:
            return None
        gt_ig0 = np.array([bool(g.get("iscrowd", 0)) or g["area"] < a_rng[0] or g["area"] > a_rng[1] for g in gts], bool)
        gtind = np.argsort(gt_ig0, kind="mergesort")            # ignored ground truths last
        gts = [gts[i] for i in gtind]
        gt_ig = gt_ig0[gtind]
        dts = dts[:max_det]                                     # already sorted by score, at most maxDets[-1]
        iscrowd = np.array([bool(g.get("iscrowd", 0)) for g in gts], bool)
        ious = ious[:len(dts)][:, gtind] if len(ious) else ious
        T, G, D = len(self.params.iouThrs), len(gts), len(dts)
        gtm, dtm, dt_ig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D), bool)
        if G and D:
            for ti, t in enumerate(self.params.iouThrs):
                for di in range(D):
                    iou, m = min(t, 1 - 1e-10), -1
                    row = ious[di]
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not iscrowd[gi]:
                            continue                          # already matched, and not a crowd
                        if m > -1 and not gt_ig[m] and gt_ig[gi]:
                            break                             # a regular match exists: stop at the ignored ones
                        if row[gi] < iou:
                            continue
                        iou, m = row[gi], gi                  # best so far (the last one among equals)
                    if m == -1:
                        continue
                    dt_ig[ti, di] = gt_ig[m]
                    dtm[ti, di] = gts[m]["id"]
                    gtm[ti, m] = dts[di]["id"]
        out_of_range = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dts], bool).reshape(1, D)
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(out_of_range, T, 0)))
        return {"dtMatches": dtm, "dtScores": np.array([d["score"] for d in dts], np.float64), "gtIgnore": gt_ig,
                "dtIgnore": dt_ig}

    def evaluate(self):
        p = self.params
        p.maxDets = sorted(p.maxDets)
        max_det = p.maxDets[-1]
        self._results = {}
        for img_id in p.imgIds:
            for cat_id in p.catIds:
                gts = self.cocoGt.img_cat_to_anns.get((img_id, cat_id), [])
                dts = self.cocoDt.img_cat_to_anns.get((img_id, cat_id), [])
                if not gts and not dts:
                    continue
                order = np.argsort([-d["score"] for d in dts], kind="mergesort")
                dts = [dts[i] for i in order[:max_det]]
                ious = bbox_iou([d["bbox"] for d in dts], [g["bbox"] for g in gts], [g.get("iscrowd", 0) for g in gts])
                for ai, a_rng in enumerate(p.areaRng):
                    self._results[(cat_id, ai, img_id)] = self._evaluate_img(img_id, cat_id, ious, gts, dts, a_rng, max_det)

    # ---- over the data set -------------------------------------------------------------------------------------
    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        eps = np.spacing(1)
        for k, cat_id in enumerate(p.catIds):
            for a in range(A):
                E = [self._results.get((cat_id, a, i)) for i in p.imgIds]
                E = [e for e in E if e is not None]
                if not E:
                    continue
                for m, max_det in enumerate(p.maxDets):
                    dt_scores = np.concatenate([e["dtScores"][:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind="mergesort")
                    dt_sorted = dt_scores[inds]
                    dtm = np.concatenate([e["dtMatches"][:, :max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, :max_det] for e in E], axis=1)[:, inds]
                    gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                    npig = np.count_nonzero(gt_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum, fp_sum = np.cumsum(tps, axis=1).astype(np.float64), np.cumsum(fps, axis=1).astype(np.float64)
                    for t in range(T):
                        tp, fp = tp_sum[t], fp_sum[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + eps)
                        q, ss = np.zeros((R,)), np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = np.maximum.accumulate(pr[::-1])[::-1]       # monotonically non-increasing from the right
                        idx = np.searchsorted(rc, p.recThrs, side="left")
                        ok = idx < nd
                        q[ok], ss[ok] = pr[idx[ok]], dt_sorted[idx[ok]]
                        precision[t, :, k, a, m], scores[t, :, k, a, m] = q, ss
        self.eval = {"precision": precision, "recall": recall, "scores": scores, "counts": [T, R, K, A, M]}

    def _summarize(self, ap=1, iou_thr=None, area="all", max_dets=100):
        p = self.params
        aind = [i for i, lbl in enumerate(p.areaRngLbl) if lbl == area]
        mind = [i for i, m in enumerate(p.maxDets) if m == max_dets]
        s = self.eval["precision"] if ap == 1 else self.eval["recall"]
        if iou_thr is not None:
            s = s[np.where(np.isclose(iou_thr, p.iouThrs))[0]]
        s = s[:, :, :, aind, mind] if ap == 1 else s[:, :, aind, mind]
        return float(np.mean(s[s > -1])) if (s > -1).any() else -1.0

    def summarize(self, print_fn=None):
        md = self.params.maxDets
        spec = [(1, None, "all", md[2]), (1, .5, "all", md[2]), (1, .75, "all", md[2]), (1, None, "small", md[2]),
                (1, None, "medium", md[2]), (1, None, "large", md[2]), (0, None, "all", md[0]), (0, None, "all", md[1]),
                (0, None, "all", md[2]), (0, None, "small", md[2]), (0, None, "medium", md[2]), (0, None, "large", md[2])]
        self.stats = np.array([self._summarize(*s) for s in spec])
        if print_fn is not None:
            for (ap, thr, area, m), v in zip(spec, self.stats):
                print_fn(" {:<18} {} @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}".format(
                    "Average Precision" if ap else "Average Recall", "(AP)" if ap else "(AR)",
                    "0.50:0.95" if thr is None else "{:0.2f}".format(thr), area, m, v))
        return self.stats
