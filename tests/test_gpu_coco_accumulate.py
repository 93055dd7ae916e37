"""§8(f)-2: COCOEvaluator.accumulate_results (eval/coco_evaluator.py:95-134) — the GPU accumulation step is
bit-exact against the numpy restatement; the class remap follows the sorted category names."""
import numpy as np
import pytest
import torch

import oracle as o

CATS = [{"id": 18, "name": "dog"}, {"id": 1, "name": "person"}, {"id": 44, "name": "bottle"}, {"id": 3, "name": "car"}]


def test_oracle_accumulate_hand_example():
    # one image 480x640 resized by 1.0 into a 640x640 canvas: scale (1, 1) / (640, 640)
    boxes = np.array([[[0.1, 0.2, 0.5, 0.9], [0.0, 0.0, 0.0, 0.0]]], np.float32)
    bb, cc = o.coco_accumulate(boxes, np.array([[2, 0]]), np.array([1]), np.array([[1.0, 1.0]], np.float32), (640, 640),
                               class_lut=[44, 3, 18, 1])
    assert bb[0, 0].tolist() == [64, 128, 320 - 64, 576 - 128] and cc[0].tolist() == [18, -1]
    assert bb[0, 1].tolist() == [0, 0, 0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("remap,rescale", [(True, True), (False, True), (True, False)])
def test_accumulate_matches_oracle(cuda, remap, rescale):
    from retinanet.eval import COCOEvaluator
    rng = np.random.default_rng(5)
    B, D = 5, 100
    x1y1 = rng.uniform(0, 0.8, (B, D, 2)).astype(np.float32)
    wh = rng.uniform(0.01, 0.4, (B, D, 2)).astype(np.float32)
    boxes = np.concatenate([x1y1, np.minimum(x1y1 + wh, 1.0)], -1).astype(np.float32)
    classes = rng.integers(0, 4, (B, D)).astype(np.int32)
    scores = rng.uniform(0.05, 1, (B, D)).astype(np.float32)
    valid = np.array([100, 0, 37, 1, 99], np.int32)
    scale = rng.uniform(0.3, 1.6, (B, 2)).astype(np.float32)
    ev = COCOEvaluator([640, 640], categories=CATS, remap_class_ids=remap)
    assert ev._lut == [44, 3, 18, 1]                       # bottle, car, dog, person
    ev.accumulate_results({"image_id": np.arange(B) + 7, "resize_scale": scale,
                           "detections": {"boxes": torch.from_numpy(boxes).to(cuda), "scores": torch.from_numpy(scores).to(cuda),
                                          "classes": torch.from_numpy(classes).to(cuda),
                                          "valid_detections": torch.from_numpy(valid).to(cuda)}},
                          rescale_detections=rescale)
    wb, wc = o.coco_accumulate(boxes, classes, valid, scale, (640, 640), class_lut=ev._lut if remap else None, rescale=rescale)
    got = ev.processed_detections
    assert len(got) == int(valid.sum())
    k = 0
    for i in range(B):
        for d in range(valid[i]):
            r = got[k]; k += 1
            assert r["image_id"] == i + 7 and r["bbox"] == wb[i, d].tolist() and r["category_id"] == int(wc[i, d])
            assert r["score"] == float(scores[i, d])
