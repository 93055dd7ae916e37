"""CPU: pin the oracle.  The reference has no tests or golden vectors, so the pins
are the hand-derived known answers of SURVEY §8(a)/(c), independent re-implementations, and
the committed fixtures under tests/golden/."""
import math
import os

import numpy as np
import pytest
import torch

import oracle as o

AREAS = [1024.0, 4096.0, 16384.0, 65536.0, 262144.0]
RATIOS = [0.5, 1.0, 2.0]
SCALES = [1, 1.2599210498948732, 1.5874010519681994]
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_anchor_known_rows_640():
    a = o.generate_anchors(640, 640, 3, 7, AREAS, RATIOS, SCALES)
    assert a.shape == (76725, 4)
    assert o.anchor_boundaries(640, 640, 3, 7, 9) == [0, 57600, 72000, 75600, 76500, 76725]
    np.testing.assert_array_equal(a[0], np.float32([4, 4, 22.627417, 45.254833]))
    np.testing.assert_array_equal(a[3], np.float32([4, 4, 32, 32]))
    np.testing.assert_array_equal(a[9], np.float32([12, 4, 22.627417, 45.254833]))
    np.testing.assert_array_equal(a[-1], np.float32([576, 576, 1149.4011, 574.70056]))


@pytest.mark.parametrize("size,n", [(1024, 196416), (1280, 306900), (896, 150381)])
def test_anchor_counts(size, n):
    assert o.anchor_boundaries(size, size, 3, 7, 9)[-1] == n


def test_focal_known_answers():
    x = [0, 0, -4.59511985013459, -4.59511985013459, 2, 2]
    y = [0, 1, 0, 1, 1, 0]
    want = [0.18379840190035, 0.06126613396678, 7.537751890126e-06, 1.13406640399704, 1.305953866395e-03,
            1.31864487750843]
    np.testing.assert_allclose(o.focal_loss_elem(x, y, 0.25, 1.5, 0.0), want, rtol=1e-9)


def test_huber_known_answers():
    np.testing.assert_allclose(o.huber_elem([0.05, 0.1, 0.3, -1.0], 0.1), [1.25e-3, 5e-3, 2.5e-2, 9.5e-2], rtol=1e-12)


def test_cosine_lr_known_answers():
    f = lambda s: o.cosine_decay_with_warmup(s, 0.32, 0.008, 500, 16875, 1e-4)
    assert f(0) == pytest.approx(0.008)
    assert f(200) == pytest.approx(0.1328)
    assert f(499) == pytest.approx(0.319376)
    assert f(500) == pytest.approx(0.31926448651596, rel=1e-12)
    assert f(8000) == pytest.approx(0.16576977293559, rel=1e-12)
    assert f(16375) == pytest.approx(3.2e-05) and f(16874) == pytest.approx(3.2e-05)


def test_rn_math_within_2ulp_of_libm():
    x = np.linspace(-30, 30, 200001).astype(np.float32)
    for fn, ref, tol in ((o.expf, np.exp, 2.0), (o.sigmoidf, lambda v: 1 / (1 + np.exp(-v)), 3.0)):
        got = fn(x).astype(np.float64)
        want = ref(x.astype(np.float64))
        ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
        assert np.max(np.abs(got - want) / ulp) <= tol
    xl = np.exp(np.linspace(-40, 40, 200001)).astype(np.float32)
    got = o.logf(xl).astype(np.float64)
    want = np.log(xl.astype(np.float64))
    ulp = np.maximum(np.spacing(np.abs(want).astype(np.float32)).astype(np.float64), 1e-45)
    assert np.max(np.abs(got - want) / ulp) <= 2.0


def test_focal_grad_matches_torch_autograd():
    rng = np.random.default_rng(0)
    x = rng.normal(-2, 3, size=4096)
    y = (rng.uniform(size=4096) < 0.1).astype(np.float64)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yt = torch.tensor(y)
    p = torch.sigmoid(xt)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(xt, yt, reduction="none")
    pt = torch.where(yt == 1, p, 1 - p)
    at = torch.where(yt == 1, torch.tensor(0.25, dtype=torch.float64), torch.tensor(0.75, dtype=torch.float64))
    loss = (at * (1 - pt) ** 1.5 * ce).sum()
    loss.backward()
    np.testing.assert_allclose(o.focal_loss_elem(x, y, 0.25, 1.5, 0.0).sum(), loss.item(), rtol=1e-12)
    np.testing.assert_allclose(o.focal_loss_grad_elem(x, y, 0.25, 1.5, 0.0), xt.grad.numpy(), rtol=1e-9, atol=1e-14)


def test_match_edge_cases():
    an = o.generate_anchors(64, 64, 3, 4, AREAS[:2], RATIOS, SCALES)
    # no GT -> everything unmatched
    m = o.match_anchor_boxes(an, np.zeros([0, 4], np.float32))
    assert (m == -1).all()
    # a GT far outside the image overlaps nothing: argmax of zeros force-matches anchor 0
    gt = np.float32([[1000, 1000, 4, 4], [20, 20, 32, 32]])
    m = o.match_anchor_boxes(an, gt)
    assert m[0] == 0
    assert (m == 1).sum() >= 1
    # two identical GTs: the anchor argmax collides, the lowest GT index wins
    gt = np.float32([[20, 20, 32, 32], [20, 20, 32, 32]])
    m = o.match_anchor_boxes(an, gt)
    iou = o.compute_iou_pairwise(gt, an)
    assert m[iou[0].argmax()] == 0
    assert not (m == 1).any()  # argmax over GT axis also returns the first maximum


def _brute_nms(boxes, scores, max_out, thr, sthr):
    order = sorted([i for i in range(len(scores)) if scores[i] > sthr], key=lambda i: (-scores[i], i))
    keep = []
    for i in order:
        if len(keep) == max_out:
            break
        ok = True
        for j in keep:
            bi, bj = boxes[i], boxes[j]
            ai = (bi[2] - bi[0]) * (bi[3] - bi[1])
            aj = (bj[2] - bj[0]) * (bj[3] - bj[1])
            if ai <= 0 or aj <= 0:
                continue
            ih = max(np.float32(min(bi[2], bj[2]) - max(bi[0], bj[0])), np.float32(0))
            iw = max(np.float32(min(bi[3], bj[3]) - max(bi[1], bj[1])), np.float32(0))
            inter = np.float32(ih * iw)
            if np.float32(inter / np.float32(np.float32(ai + aj) - inter)) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def test_hard_nms_matches_brute_force():
    rng = np.random.default_rng(3)
    for trial in range(5):
        n = 300
        c = rng.uniform(0.2, 0.8, (n, 2)).astype(np.float32)
        wh = rng.uniform(0.05, 0.3, (n, 2)).astype(np.float32)
        boxes = np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        scores[::7] = scores[3]  # duplicate scores -> index tie-break
        idx, sc, nv = o.nms_v5(boxes, scores, 50, 0.5, 0.05, 0.0)
        assert list(idx[:nv]) == _brute_nms(boxes, scores, 50, 0.5, 0.05)
        assert (idx[nv:] == 0).all() and (sc[nv:] == 0).all()


def test_soft_nms_properties():
    rng = np.random.default_rng(4)
    c = rng.uniform(0.3, 0.7, (200, 2)).astype(np.float32)
    wh = rng.uniform(0.1, 0.3, (200, 2)).astype(np.float32)
    boxes = np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)
    scores = rng.uniform(0.1, 1, 200).astype(np.float32)
    idx, sc, nv = o.nms_v5(boxes, scores, 100, 1.0, 0.05, 0.25)
    assert nv > 0 and idx[0] == int(np.lexsort((np.arange(200), -scores))[0])
    assert sc[0] == scores[idx[0]]
    assert (sc[:nv] <= scores[idx[:nv]]).all() and (sc[:nv] > 0.05).all()
    assert len(set(idx[:nv].tolist())) == nv
    # sigma -> 0+ limit is not hard NMS, but a huge sigma never decays: selection = score order
    idx2, sc2, nv2 = o.nms_v5(boxes, scores, 20, 1.0, 0.05, 1e9)
    np.testing.assert_array_equal(idx2, np.lexsort((np.arange(200), -scores))[:20])


def test_golden_fixtures_reproduce():
    """tests/golden/*.npz were generated by make_golden.py with this oracle; regenerate in
    memory and require bit-identical outputs (guards the oracle against silent edits)."""
    import make_golden
    for name, arrays in make_golden.generate().items():
        with np.load(os.path.join(GOLD, name + ".npz")) as z:
            assert sorted(z.files) == sorted(arrays)
            for k in z.files:
                np.testing.assert_array_equal(z[k], arrays[k], err_msg=f"{name}:{k}")


def test_prepare_image_matches_torch_bilinear():
    """Independent pin of the TF2 bilinear restatement (half-pixel centres == torch align_corners=False)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    for (h, w, t) in ((37, 53, 128), (480, 640, 640), (700, 500, 320)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.float32)
        out, sc = o.prepare_image(img, t, t, [0, 0, 0], [1, 1, 1], 1.0)
        r = min(np.float32(t) / np.float32(h), np.float32(t) / np.float32(w))
        sh, sw = int(np.round(np.float32(h) * r)), int(np.round(np.float32(w) * r))
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(sh, sw), mode="bilinear",
                            align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        np.testing.assert_allclose(out[:sh, :sw], ref, rtol=0, atol=1e-3)
        assert (out[sh:] == 0).all() and (out[:, sw:] == 0).all() and max(sh, sw) == t
        np.testing.assert_allclose(sc, [sh / h, sw / w], rtol=1e-6)
