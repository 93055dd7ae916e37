"""GPU parity: a9 focal + Huber loss forward/backward within 1e-5 of the float64 oracle."""
import os

import numpy as np
import pytest
import torch

import oracle as o

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-5  # north_star: "focal/box-loss within 1e-5 fp32"


def _run(cuda, logits, box_preds, cls_t, box_t, npos_sum, K, splits, **kw):
    from retinanet.cfg import default_params
    from retinanet.losses import RetinaNetLoss
    p = default_params()
    for k, v in kw.items():
        if k in ("alpha", "gamma", "label_smoothing"):
            p.loss.focal_loss[k] = v
        elif k == "delta":
            p.loss.smooth_l1_loss.delta = v
    loss = RetinaNetLoss(K, p.loss)
    B = logits.shape[0]
    preds = {"class-predictions": {}, "box-predictions": {}}
    off = 0
    for i, n in enumerate(splits):
        preds["class-predictions"][str(3 + i)] = torch.from_numpy(logits[:, off:off + n].copy()).to(cuda)
        preds["box-predictions"][str(3 + i)] = torch.from_numpy(box_preds[:, off:off + n].copy()).to(cuda)
        off += n
    targets = {"num-positives": torch.tensor([npos_sum] + [0.0] * (B - 1), device=cuda),
               "_flat": {"class-targets": torch.from_numpy(cls_t).to(cuda), "box-targets": torch.from_numpy(box_t).to(cuda)}}
    out = loss(targets, preds)
    torch.cuda.synchronize()
    dl = torch.cat([loss.grads["class-predictions"][str(3 + i)] for i in range(len(splits))], dim=1).cpu().numpy()
    db = torch.cat([loss.grads["box-predictions"][str(3 + i)] for i in range(len(splits))], dim=1).cpu().numpy()
    return {k: (v.item() if torch.is_tensor(v) else v) for k, v in out.items()}, dl, db


def test_loss_golden(cuda):
    with np.load(os.path.join(GOLD, "loss_128.npz")) as z:
        A = z["logits"].shape[1]
        out, dl, db = _run(cuda, z["logits"], z["box_preds"], z["cls_t"], z["box_t"], float(z["normalizer"]) - 1.0, 6,
                           [2304, 576, 144, 36, 9])
        assert sum([2304, 576, 144, 36, 9]) == A
        np.testing.assert_allclose([out["box-loss"], out["class-loss"], out["weighted-loss"]], z["losses"], rtol=RTOL)
        assert out["num-anchors-matched"] == pytest.approx(float(z["normalizer"]))
        scale = np.abs(z["dlogits"]).max()
        np.testing.assert_allclose(dl, z["dlogits"], rtol=RTOL, atol=RTOL * scale)
        np.testing.assert_allclose(db, z["dbox"], rtol=RTOL, atol=RTOL * np.abs(z["dbox"]).max())


@pytest.mark.parametrize("K,ls,gamma", [(80, 0.0, 1.5), (80, 0.1, 2.0), (7, 0.0, 1.5)])
def test_loss_vs_oracle_640(cuda, K, ls, gamma):
    rng = np.random.default_rng(K)
    B, A = 2, 76725
    logits = rng.normal(-4.595, 1.0, (B, A, K)).astype(np.float32)
    logits[0, :50] = rng.normal(0, 6, (50, K))      # both sigmoid tails
    box_preds = rng.normal(0, 0.3, (B, A, 4)).astype(np.float32)
    cls_t = np.full([B, A], -1.0, np.float32)
    pos = rng.choice(A, 300, replace=False)
    cls_t[0, pos] = rng.integers(0, K, 300)
    cls_t[1, pos[:50]] = -2.0
    box_t = np.zeros([B, A, 4], np.float32)
    box_t[0, pos] = rng.normal(0, 0.5, (300, 4))
    box_t[0, pos[0], 1] = 0.0                       # zero coordinate -> weight 0 (loss_impl.py:96)
    out, dl, db = _run(cuda, logits, box_preds, cls_t, box_t, 300.0, K, [57600, 14400, 3600, 900, 225],
                       label_smoothing=ls, gamma=gamma)
    ref, rdl, rdb = o.retinanet_loss(logits, box_preds, cls_t, box_t, 301.0, K, gamma=gamma, label_smoothing=ls)
    np.testing.assert_allclose(out["class-loss"], ref["class-loss"], rtol=RTOL)
    np.testing.assert_allclose(out["box-loss"], ref["box-loss"], rtol=RTOL)
    np.testing.assert_allclose(out["weighted-loss"], ref["weighted-loss"], rtol=RTOL)
    np.testing.assert_allclose(dl, rdl, rtol=RTOL, atol=RTOL * np.abs(rdl).max())
    np.testing.assert_allclose(db, rdb, rtol=RTOL, atol=RTOL * np.abs(rdb).max())


def test_fresh_head_all_background(cuda):
    """an all-background image at the class-bias init gives
    class-loss * normaliser ~= A*K * 7.5378e-06 per image."""
    B, A, K = 1, 76725, 80
    logits = np.full([B, A, K], -4.59511985013459, np.float32)
    out, dl, db = _run(cuda, logits, np.zeros([B, A, 4], np.float32), np.full([B, A], -1.0, np.float32),
                       np.zeros([B, A, 4], np.float32), 0.0, K, [57600, 14400, 3600, 900, 225])
    assert out["class-loss"] == pytest.approx(A * K * 7.537751890126e-06, rel=2e-5)
    assert out["box-loss"] == 0.0 and (db == 0).all()


def test_loss_bf16_gradient_outputs_match_the_cast(cuda):
    """rn_retinanet_loss_fwd_bwd_bf16 writes the gradients as bf16 into channel-padded NHWC tensors (the dy tensors of
    the prediction convs): same losses, and bit for bit the round-to-nearest-even cast of the fp32 gradients; the
    pad channels are left alone."""
    from retinanet.cfg import default_params
    from retinanet.losses import RetinaNetLoss
    K, na, B = 8, 3, 2
    sizes = [(8, 8), (4, 4), (2, 2)]
    g = torch.Generator().manual_seed(5)
    loss = RetinaNetLoss(K, default_params().loss)
    preds = {"class-predictions": {}, "box-predictions": {}}
    for i, (h, w) in enumerate(sizes):
        preds["class-predictions"][str(3 + i)] = (torch.randn((B, h, w, na * K), generator=g) * 2 - 2).to(cuda)
        preds["box-predictions"][str(3 + i)] = torch.randn((B, h, w, na * 4), generator=g).to(cuda)
    A = sum(h * w * na for h, w in sizes)
    cls_t = torch.randint(-2, K, (B, A), generator=g).float()
    box_t = torch.randn((B, A, 4), generator=g) * (cls_t >= 0).float()[..., None]
    targets = {"num-positives": torch.tensor([float((cls_t >= 0).sum()), 0.0], device=cuda),
               "_flat": {"class-targets": cls_t.to(cuda), "box-targets": box_t.to(cuda)}}
    out32 = loss(targets, preds)
    g32 = loss.grads
    cs, bs = 32, 16     # padded channel strides (24 / 12 live)
    bufs = {"class-predictions": {lv: torch.full((B, h, w, cs), 7.0, dtype=torch.bfloat16, device=cuda)
                                  for lv, (h, w) in zip("345", sizes)},
            "box-predictions": {lv: torch.full((B, h, w, bs), 7.0, dtype=torch.bfloat16, device=cuda)
                                for lv, (h, w) in zip("345", sizes)}}
    out16 = loss(targets, preds, grads_bf16=bufs)
    torch.cuda.synchronize()
    assert loss.grads is None
    for k in ("box-loss", "class-loss", "weighted-loss"):
        assert out32[k].item() == out16[k].item()
    for key, live in (("class-predictions", na * K), ("box-predictions", na * 4)):
        for lv in "345":
            got = bufs[key][lv]
            assert torch.equal(got[..., :live], g32[key][lv].to(torch.bfloat16))
            assert bool((got[..., live:] == 7.0).all())
    assert float(g32["class-predictions"]["3"].abs().max()) > 0
