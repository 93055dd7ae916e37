"""CPU: the NumPy COCOeval (retinanet/eval/cocoeval.py, the bbox branch of pycocotools 2.0.2's COCOeval which the
reference's evaluator calls, eval/coco_evaluator.py:136-157).  pycocotools is not installable here, so the restatement
is pinned by cases whose twelve statistics are worked out by hand from the published definition, by an independent
brute-force AP (other code structure: one global score order, per-threshold loop), and by invariances."""
import numpy as np
import pytest

from retinanet.eval.cocoeval import COCO, COCOeval, bbox_iou


def _gt(anns, images=(1,), cats=(1,)):
    return COCO(dataset={"images": [{"id": i} for i in images], "categories": [{"id": c, "name": str(c)} for c in cats],
                         "annotations": [dict(a, id=k + 1, iscrowd=a.get("iscrowd", 0),
                                              area=a.get("area", a["bbox"][2] * a["bbox"][3])) for k, a in enumerate(anns)]})


def _run(gt, dets):
    ev = COCOeval(gt, gt.loadRes(dets), "bbox")
    ev.evaluate()
    ev.accumulate()
    return ev.summarize()


def test_hand_computed_single_image():
    """GT A=[0,0,10,10], B=[20,20,10,10]; detections d1 = A (0.9), d2 = upper half of B (0.8, IoU exactly 0.5),
    d3 elsewhere (0.7).  IoU 0.50: TP TP FP -> precision 1 at every recall -> AP 1.  IoU 0.55..0.95: TP FP FP ->
    recall stops at 0.5: precision 1 for the 51 recall thresholds <= 0.5, 0 above -> 51/101."""
    gt = _gt([{"image_id": 1, "category_id": 1, "bbox": [0, 0, 10, 10]}, {"image_id": 1, "category_id": 1, "bbox": [20, 20, 10, 10]}])
    dets = [{"image_id": 1, "category_id": 1, "bbox": [0, 0, 10, 10], "score": 0.9},
            {"image_id": 1, "category_id": 1, "bbox": [20, 20, 10, 5], "score": 0.8},
            {"image_id": 1, "category_id": 1, "bbox": [50, 50, 10, 10], "score": 0.7}]
    s = _run(gt, dets)
    ap_hi = 51 / 101
    assert s[0] == pytest.approx((1.0 + 9 * ap_hi) / 10)        # AP@[.50:.95]
    assert s[1] == pytest.approx(1.0) and s[2] == pytest.approx(ap_hi)
    assert s[3] == pytest.approx(s[0])                          # every box is "small" (< 32^2)
    assert s[4] == -1 and s[5] == -1                            # no medium / large ground truth
    assert s[6] == pytest.approx(0.5)                           # AR@1: only d1 counts
    assert s[7] == pytest.approx((1.0 + 9 * 0.5) / 10) and s[8] == pytest.approx(s[7])
    assert s[9] == pytest.approx(s[8]) and s[10] == -1 and s[11] == -1
    # the order the detections are listed in does not matter
    np.testing.assert_allclose(_run(gt, dets[::-1]), s)


def test_crowd_and_area_ignore_rules():
    """a crowd ground truth never counts as a miss, matches any number of detections (IoU = inter / det area), and the
    detections it absorbs are ignored instead of counted as false positives; a detection on a ground truth outside
    the area range is ignored in that range."""
    gt = _gt([{"image_id": 1, "category_id": 1, "bbox": [0, 0, 10, 10]},
              {"image_id": 1, "category_id": 1, "bbox": [100, 100, 100, 100], "iscrowd": 1}])
    dets = [{"image_id": 1, "category_id": 1, "bbox": [0, 0, 10, 10], "score": 0.9},
            {"image_id": 1, "category_id": 1, "bbox": [110, 110, 20, 20], "score": 0.8},     # inside the crowd region
            {"image_id": 1, "category_id": 1, "bbox": [150, 150, 20, 20], "score": 0.7}]     # a second one
    s = _run(gt, dets)
    assert s[0] == pytest.approx(1.0) and s[8] == pytest.approx(1.0)     # one regular GT, found, nothing else counted
    assert bbox_iou([[110, 110, 20, 20]], [[100, 100, 100, 100]], [1])[0, 0] == pytest.approx(1.0)
    assert bbox_iou([[110, 110, 20, 20]], [[100, 100, 100, 100]], [0])[0, 0] == pytest.approx(400 / 10000)
    # medium range: the 10x10 ground truth is out of range -> ignored there, its detection too -> no GT -> -1
    assert s[4] == -1
    # a large ground truth and its detection only show up under "large"
    gt2 = _gt([{"image_id": 1, "category_id": 1, "bbox": [0, 0, 200, 200]}])
    s2 = _run(gt2, [{"image_id": 1, "category_id": 1, "bbox": [0, 0, 200, 200], "score": 0.5}])
    assert s2[0] == pytest.approx(1.0) and s2[5] == pytest.approx(1.0) and s2[3] == -1 and s2[11] == pytest.approx(1.0)


def test_no_detections_and_unknown_image():
    gt = _gt([{"image_id": 1, "category_id": 1, "bbox": [0, 0, 10, 10]}])
    ev = COCOeval(gt, COCO(dataset={"images": [{"id": 1}], "categories": [{"id": 1, "name": "1"}], "annotations": []}))
    ev.evaluate(); ev.accumulate()
    s = ev.summarize()
    assert s[0] == 0.0 and s[8] == 0.0
    with pytest.raises(AssertionError):
        gt.loadRes([{"image_id": 7, "category_id": 1, "bbox": [0, 0, 1, 1], "score": 0.1}])


def _brute_force_ap(gts, dets, thr):
    """AP at one IoU threshold, one category, no crowd / area rules: one global pass over the detections in score
    order, each taking the unmatched ground truth of its image with the highest IoU >= thr."""
    dets = sorted(dets, key=lambda d: -d["score"])
    taken = set()
    tp = []
    for d in dets:
        best, best_iou = None, thr
        for gi, g in enumerate(gts):
            if g["image_id"] != d["image_id"] or gi in taken:
                continue
            iou = bbox_iou([d["bbox"]], [g["bbox"]], [0])[0, 0]
            if iou >= best_iou:
                best, best_iou = gi, iou
        tp.append(best is not None)
        if best is not None:
            taken.add(best)
    tp = np.array(tp, bool)
    ctp, cfp = np.cumsum(tp), np.cumsum(~tp)
    rc, pr = ctp / len(gts), ctp / np.maximum(ctp + cfp, 1e-300)
    for i in range(len(pr) - 1, 0, -1):
        pr[i - 1] = max(pr[i - 1], pr[i])
    ap = 0.0
    for r in np.linspace(0, 1, 101):
        k = np.searchsorted(rc, r, side="left")
        ap += pr[k] if k < len(pr) else 0.0
    return ap / 101, rc[-1]


def test_against_independent_brute_force():
    rng = np.random.default_rng(7)
    gts, dets = [], []
    for img in range(1, 13):
        for _ in range(int(rng.integers(1, 6))):
            x, y, w, h = rng.uniform(0, 300), rng.uniform(0, 300), rng.uniform(40, 120), rng.uniform(40, 120)
            gts.append({"image_id": img, "category_id": 1, "bbox": [x, y, w, h]})
            if rng.uniform() < 0.8:     # a jittered detection of it
                j = rng.normal(0, 12, 4)
                dets.append({"image_id": img, "category_id": 1, "bbox": [x + j[0], y + j[1], max(w + j[2], 5), max(h + j[3], 5)],
                             "score": float(rng.uniform(0.3, 1.0))})
        for _ in range(int(rng.integers(0, 4))):   # clutter
            dets.append({"image_id": img, "category_id": 1, "bbox": [rng.uniform(0, 300), rng.uniform(0, 300), 60, 60],
                         "score": float(rng.uniform(0.0, 0.6))})
    gt = _gt(gts, images=range(1, 13))
    ev = COCOeval(gt, gt.loadRes(dets), "bbox")
    ev.evaluate(); ev.accumulate(); ev.summarize()
    P, R = ev.eval["precision"], ev.eval["recall"]
    for ti, thr in enumerate(ev.params.iouThrs):
        ap, rec = _brute_force_ap(gts, dets, thr)
        assert float(np.mean(P[ti, :, 0, 0, 2])) == pytest.approx(ap, abs=1e-12), thr
        assert R[ti, 0, 0, 2] == pytest.approx(rec), thr
    assert ev.stats[0] == pytest.approx(np.mean([_brute_force_ap(gts, dets, t)[0] for t in ev.params.iouThrs]))


def test_categories_are_evaluated_separately_and_averaged():
    gt = _gt([{"image_id": 1, "category_id": 1, "bbox": [0, 0, 50, 50]}, {"image_id": 1, "category_id": 2, "bbox": [100, 100, 50, 50]}],
             cats=(1, 2))
    dets = [{"image_id": 1, "category_id": 1, "bbox": [0, 0, 50, 50], "score": 0.9},
            {"image_id": 1, "category_id": 1, "bbox": [100, 100, 50, 50], "score": 0.8}]     # right box, wrong class
    s = _run(gt, dets)
    assert s[0] == pytest.approx(0.5) and s[8] == pytest.approx(0.5)      # class 1 perfect, class 2 never found
