"""GPU: sigmoid / exp / log of the HIP kernels against numpy's float64 libm — NOT through include/rn_math.h
(VERDICT r5 weak 1b / next 3b).

Every bit-exact test of the post-processing and target-encoding kernels compares the GPU with an oracle that is compiled
from the SAME header (include/rn_math.h, oracle/rn_oracle.c:23): a wrong coefficient there would be invisible to all of
them.  Here the expected values come from `np.exp` / `np.log` in float64 on the kernel's own float32 inputs; nothing under
`oracle/` is imported.  Tolerances (float32 ulps of the expected value): sigmoid <= 3 (1 / (1 + e^-x): exp 2 ulp, one add,
one correctly rounded divide), decode <= 4 ulp of the box's own scale (exp 2 ulp, then a multiply, a halving, an add and a
divide), box targets' log <= 2 ulp + the conditioning of log near 1 (the argument is itself a rounded float32 quotient,
reproduced here in float32).  (The loss kernels use hardware exp2 / log2 forms, not rn_math.h, and tests/test_gpu_loss.py
already compares them with a float64 numpy restatement at 1e-5.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F32 = np.float32


def _ulps(got, want64):
    """|got - want| in units of the float32 spacing at |want| (spacing floored at the smallest normal's)"""
    want32 = want64.astype(F32)
    sp = np.spacing(np.maximum(np.abs(want32), F32(1.17549435e-38))).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / sp


def _params(size, K):
    from retinanet.cfg import default_params
    p = default_params(input_size=size)
    p.architecture.head.num_classes = K
    return p


def test_sigmoid_and_decode_against_float64_libm(cuda):
    from retinanet.model.layers import TransformBoxesAndScores
    rng = np.random.default_rng(5)
    size, K, B = 256, 8, 2
    p = _params(size, K)
    tb = TransformBoxesAndScores(p)
    an = tb._anchors.boxes.cpu().numpy()
    A = an.shape[0]
    logits = rng.normal(-3.0, 3.0, (B, A, K)).astype(F32)
    logits[0, :64, 0] = np.linspace(-100.0, 100.0, 64, dtype=F32)      # both tails: 0 < sigmoid <= 1, no NaN
    logits[0, 64:72, 0] = [0.0, -0.0, 1e-8, -1e-8, 88.0, -88.0, 17.0, -17.0]
    enc = rng.normal(0.0, 0.5, (B, A, 4)).astype(F32)
    enc[1, :32, 2:] = np.linspace(-6.0, 6.0, 64, dtype=F32).reshape(32, 2)
    out = tb({"class_logits": torch.from_numpy(logits).to(cuda), "encoded_boxes": torch.from_numpy(enc).to(cuda)})
    torch.cuda.synchronize()
    scores, boxes = out["scores"].cpu().numpy(), out["boxes"].cpu().numpy()
    # (1) scores = sigmoid(logits)
    want = 1.0 / (1.0 + np.exp(-logits.astype(np.float64)))
    assert np.isfinite(scores).all() and scores.min() >= 0.0 and scores.max() <= 1.0
    body = np.abs(logits) <= 87.0          # e^-x stays a normal float32 here
    u = _ulps(scores[body], want[body])
    assert u.max() <= 3.0, u.max()
    assert (u <= 1.0).mean() > 0.95
    # beyond: 1 + e^-x overflows to +inf and the quotient is 0 where the true value is a denormal (< 1.2e-38); the upper
    # tail is exactly 1
    assert np.abs(scores[~body].astype(np.float64) - want[~body]).max() < 1.2e-38
    # (2) boxes: [xy - wh/2, xy + wh/2] / [H, W, H, W] with xy = t_xy * a_wh + a_xy, wh = exp(t_wh) * a_wh
    # (postprocessing_ops.py:87-117); error measured against the scale of the box (its largest |coordinate term|)
    t, a = enc.astype(np.float64), an.astype(np.float64)[None]
    xy = t[..., :2] * a[..., 2:] + a[..., :2]
    wh = np.exp(t[..., 2:]) * a[..., 2:]
    want_b = np.concatenate([xy - wh / 2, xy + wh / 2], axis=-1) / float(size)
    scale = (np.abs(xy) + wh / 2).max(axis=-1, keepdims=True) / float(size)
    err = np.abs(boxes.astype(np.float64) - want_b) / np.spacing(scale.astype(F32)).astype(np.float64)
    assert err.max() <= 4.0, err.max()


def test_box_target_log_against_float64_libm(cuda):
    """label_encoder.py:57-71: t_wh = log(max(g_wh, 1e-8) / a_wh) for the matched anchors"""
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    size = 640
    p = default_params(input_size=size)
    enc = LabelEncoder(p, device=cuda)
    rng = np.random.default_rng(9)
    G = 48
    c = rng.uniform(40, 600, (G, 2))
    wh = np.exp(rng.uniform(np.log(6), np.log(500), (G, 2)))
    gb = np.concatenate([c, wh], axis=1).astype(F32)[None]          # [cx, cy, w, h]
    gc = rng.integers(0, 80, (1, G)).astype(F32)
    t = enc.encode_batch(torch.from_numpy(gb), torch.from_numpy(gc), torch.tensor([G], dtype=torch.int32))
    torch.cuda.synchronize()
    m = t["_flat"]["matches"][0].cpu().numpy()
    bt = t["_flat"]["box-targets"][0].cpu().numpy()
    an = enc.anchors.boxes.cpu().numpy()
    pos = np.where(m >= 0)[0]
    assert len(pos) > 300
    g = gb[0][m[pos]]
    a = an[pos]
    # the quotient as the kernel forms it (float32), the log in float64
    ratio32 = (np.maximum(g[:, 2:], F32(1e-8)) / a[:, 2:]).astype(F32)
    want_wh = np.log(ratio32.astype(np.float64))
    got_wh = bt[pos, 2:]
    # 2 ulp of the result; near log(1) = 0 the result's ulp shrinks while the argument's rounding does not: allow the
    # half-ulp of the argument (6e-8 relative) on top
    tol = 2.0 * np.spacing(np.abs(want_wh).astype(F32)).astype(np.float64) + 6e-8
    assert (np.abs(got_wh.astype(np.float64) - want_wh) <= tol).all()
    want_xy = ((g[:, :2] - a[:, :2]) / a[:, 2:]).astype(F32)       # one subtract, one correctly rounded divide: exact
    np.testing.assert_array_equal(bt[pos, :2], want_xy)
    assert (bt[m < 0] == 0).all()
