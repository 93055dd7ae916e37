"""GPU parity for SURVEY §8 row a18: the depthwise conv and squeeze-excite kernels against
torch-CPU float32, and the EfficientNet-B3 + separable-conv FPN/head forward pass against the
CPU restatement (oracle/model_ref.py, efficientnet.py:222-265, 291-482, 566-586, 783-855)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from model_ref import RefModel

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _same(x, k, s):
    H, W = x.shape[2], x.shape[3]
    ph = max((-(-H // s) - 1) * s + k - H, 0)
    pw = max((-(-W // s) - 1) * s + k - W, 0)
    return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)), ph // 2, pw // 2


@pytest.mark.parametrize("N,H,W,C,k,s,act", [
    (2, 32, 32, 144, 3, 1, "swish"), (2, 40, 40, 192, 5, 2, "swish"), (1, 20, 20, 816, 5, 1, "swish"),
    (2, 16, 16, 160, 3, 1, None), (2, 13, 11, 48, 1, 1, None), (1, 33, 31, 96, 3, 2, "relu")])
def test_depthwise_conv(cuda, N, H, W, C, k, s, act):
    from retinanet import _C
    lib = _C.lib()
    g = torch.Generator().manual_seed(C + k)
    x = _bf(torch.randn((N, H, W, C), generator=g))
    w = torch.randn((k, k, C, 1), generator=g) * (1.0 / k)
    scale = torch.rand(C, generator=g) + 0.5
    shift = torch.randn(C, generator=g) * 0.1
    xp, pt, pl = _same(x.permute(0, 3, 1, 2), k, s)
    want = F.conv2d(xp, _bf(w).permute(2, 3, 0, 1).contiguous(), None, stride=s, groups=C)
    want = want * scale[None, :, None, None] + shift[None, :, None, None]
    if act == "swish":
        want = want * torch.sigmoid(want)
    elif act == "relu":
        want = F.relu(want)
    want = want.permute(0, 2, 3, 1)
    Ho, Wo = want.shape[1], want.shape[2]
    st = _C.current_stream()
    xd = x.to(cuda, torch.bfloat16).contiguous()
    wd = torch.empty((k * k, C), dtype=torch.bfloat16, device=cuda)
    wf = w.to(cuda).contiguous()
    _C.check(lib.rn_pack_depthwise_weight(_C.ptr(wf), k, C, _C.ptr(wd), st), "pack")
    y = torch.empty((N, Ho, Wo, C), dtype=torch.bfloat16, device=cuda)
    sc, sh = scale.to(cuda), shift.to(cuda)
    p = _C.DwProblem()
    p.k, p.stride, p.pad_top, p.pad_left, p.act, p.num_segments = k, s, pt, pl, _C.ACT_IDS[act], 1
    sg = p.seg[0]
    sg.x, sg.w, sg.y, sg.scale, sg.shift = xd.data_ptr(), wd.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
    sg.N, sg.H, sg.W, sg.C, sg.Ho, sg.Wo = N, H, W, C, Ho, Wo
    _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(ctypes.byref(p), st), "dw")
    torch.cuda.synchronize()
    got = y.float().cpu()
    # fp32 accumulation in a different order + one bf16 rounding of the output
    tol = 2.0 ** -7 * want.abs().clamp_min(1.0)
    assert ((got - want).abs() <= tol).all(), (got - want).abs().max().item()


@pytest.mark.parametrize("N,HW,C,se", [(2, 40 * 40, 144, 6), (3, 20 * 20, 816, 34), (1, 7 * 5, 2304, 96),
                                         (2, 1, 40, 10)])
def test_squeeze_excite(cuda, N, HW, C, se):
    from retinanet import _C
    lib = _C.lib()
    g = torch.Generator().manual_seed(C)
    x = _bf(torch.randn((N, HW, C), generator=g) + 0.3)
    w1 = _bf(torch.randn((se, C), generator=g) * (2.0 / C) ** 0.5)
    b1 = torch.randn(se, generator=g) * 0.1
    w2 = _bf(torch.randn((C, se), generator=g) * (2.0 / se) ** 0.5)
    b2 = torch.randn(C, generator=g) * 0.1
    # every Keras layer output under the mixed policy is a 16-bit tensor: round where SE.call
    # (efficientnet.py:252-265) materialises one
    pooled = _bf(x.mean(dim=1))
    h = _bf(pooled @ w1.t() + b1)
    h = _bf(h * torch.sigmoid(h))
    gate = _bf(torch.sigmoid(_bf(h @ w2.t() + b2)))
    want = x * gate[:, None, :]
    xd = x.to(cuda, torch.bfloat16).contiguous()
    ws = torch.empty(lib.rn_se_workspace_bytes(N, C), dtype=torch.uint8, device=cuda)
    args = [t.to(cuda).contiguous() for t in (w1.to(torch.bfloat16), b1, w2.to(torch.bfloat16), b2)]
    _C.check(lib.rn_squeeze_excite_inplace(_C.ptr(xd), N, HW, C, _C.ptr(args[0]), _C.ptr(args[1]), _C.ptr(args[2]),
                                           _C.ptr(args[3]), se, _C.ptr(ws), ws.numel(), _C.current_stream()), "se")
    torch.cuda.synchronize()
    got = xd.float().cpu()
    # a 1-ulp flip of an intermediate bf16 rounding moves the gate by <= 2^-8 relative
    tol = 2.0 ** -6 * want.abs().clamp_min(0.05)
    assert ((got - want).abs() <= tol).all(), (got - want).abs().max().item()
    assert ((got - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6).float().mean() > 0.97


def _randomize(model, seed):
    g = torch.Generator().manual_seed(seed)
    for k, v in model.variables.items():
        if k.endswith("/gamma"):
            v.copy_((torch.rand(v.shape, generator=g) * 0.5 + 0.75).to(v.device))
        elif k.endswith("/beta") or k.endswith("/moving_mean"):
            v.copy_((torch.randn(v.shape, generator=g) * 0.1).to(v.device))
        elif k.endswith("/moving_variance"):
            v.copy_((torch.rand(v.shape, generator=g) * 0.5 + 0.75).to(v.device))
        elif k.endswith("/bias") and "prediction" not in k:
            v.copy_((torch.randn(v.shape, generator=g) * 0.05).to(v.device))
    model._refresh()


@pytest.mark.parametrize("size,B", [(256, 2), (384, 1)])
def test_efficientnet_b3_forward_matches_cpu_restatement(cuda, size, B):
    from retinanet.cfg import efficientnet_params
    from retinanet.model import ModelBuilder
    p = efficientnet_params("efficientnet-b3", input_size=size)
    model = ModelBuilder(p, "val", device=cuda)()
    _randomize(model, 3)
    g = torch.Generator().manual_seed(1337)
    images = torch.randn((B, size, size, 3), generator=g)
    preds = model(images.to(cuda), training=False)
    torch.cuda.synchronize()
    ref = RefModel(p, model.variables, emulate_bf16=True)(images)
    for key in ("box-predictions", "class-predictions"):
        for lv in ("3", "4", "5", "6", "7"):
            got = preds[key][lv].float().cpu()
            want = ref[key][lv]
            assert got.shape == want.shape
            scale = want.abs().max().item()
            err = (got - want).abs()
            # bf16 activations through ~110 layers: 1-ulp flips propagate; bound max and mean
            assert err.max().item() <= 0.08 * scale + 1e-3, (key, lv, err.max().item(), scale)
            assert err.mean().item() <= 0.01 * scale + 1e-4, (key, lv, err.mean().item(), scale)


def test_efficientnet_serving_soft_nms(cuda):
    """BASELINE config 4 end to end (PerClassSoftNMS, sigma 0.5): HIP post-process of the HIP head
    outputs is bit-exact against the oracle post-process of the same head outputs."""
    import oracle as o
    from retinanet.cfg import efficientnet_params
    from retinanet.model import ModelBuilder
    p = efficientnet_params("efficientnet-b3", input_size=256)
    assert p.inference.mode == "PerClassSoftNMS"
    p.inference.score_threshold = 0.005
    b = ModelBuilder(p, "val", device=cuda)
    model = b()
    _randomize(model, 5)
    infer = b.add_post_processing_stage(model)
    images = torch.randn((2, 256, 256, 3), generator=torch.Generator().manual_seed(7)).to(cuda)
    out = {k: v.cpu().numpy().copy() for k, v in infer(images).items()}
    preds = model(images)
    torch.cuda.synchronize()
    logits = np.concatenate([preds["class-predictions"][l].cpu().numpy().reshape(2, -1, 80) for l in "34567"], axis=1)
    enc = np.concatenate([preds["box-predictions"][l].cpu().numpy().reshape(2, -1, 4) for l in "34567"], axis=1)
    an = o.generate_anchors(256, 256, 3, 7, p.anchor_params.areas, p.anchor_params.aspect_ratios, p.anchor_params.scales)
    wb, ws, wc, wv = o.postprocess(logits, enc, an, 256, 256, score_threshold=0.005,
                                   sigma=float(p.inference.soft_nms_sigma))
    assert wv.min() > 0
    np.testing.assert_array_equal(out["valid_detections"], wv)
    np.testing.assert_array_equal(out["classes"], wc)
    np.testing.assert_array_equal(out["scores"], ws)
    np.testing.assert_array_equal(out["boxes"], wb)
