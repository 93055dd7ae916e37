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
    # bf16 tensors between the layers (include/rnet_hip.h): DepthwiseConv2D output, BatchNorm output in front of swish
    want = _bf(F.conv2d(xp, _bf(w).permute(2, 3, 0, 1).contiguous(), None, stride=s, groups=C))
    want = want * scale[None, :, None, None] + shift[None, :, None, None]
    if act == "swish":
        want = _bf(want)
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


@pytest.mark.parametrize("N,H,W,C,k,s", [(2, 16, 16, 144, 3, 1), (2, 20, 20, 96, 5, 2), (1, 9, 7, 64, 5, 1),
                                         (2, 12, 12, 48, 3, 2), (1, 8, 8, 160, 1, 1),
                                         # the weight gradient's workgroups cover min(C / channels-per-thread, 64) channel
                                         # groups: one group, a whole odd-sized pixel, several slabs with a short last one
                                         (1, 6, 6, 8, 3, 1), (2, 14, 14, 40, 3, 1), (1, 10, 10, 816, 5, 1),
                                         (1, 6, 6, 1392, 5, 2), (1, 6, 6, 2304, 3, 1), (1, 8, 8, 288, 3, 2)])
def test_depthwise_backward(cuda, N, H, W, C, k, s):
    """dgrad = the forward kernel on (zero-upsampled) dy with the tap-reversed filter; wgrad = the two-stage
    reduction kernel; both against torch autograd on the same bf16-rounded operands."""
    from retinanet import _C
    lib = _C.lib()
    g = torch.Generator().manual_seed(C * k + s)
    x = _bf(torch.randn((N, H, W, C), generator=g))
    w = torch.randn((k, k, C, 1), generator=g) * (1.0 / k)
    xp, pt, pl = _same(x.permute(0, 3, 1, 2), k, s)
    xa = x.double().requires_grad_(True)
    wa = _bf(w).double().requires_grad_(True)
    xpa, _, _ = _same(xa.permute(0, 3, 1, 2), k, s)
    ya = F.conv2d(xpa, wa.permute(2, 3, 0, 1), None, stride=s, groups=C).permute(0, 2, 3, 1)
    Ho, Wo = ya.shape[1], ya.shape[2]
    dy = _bf(torch.randn((N, Ho, Wo, C), generator=g))
    (ya * dy.double()).sum().backward()
    st = _C.current_stream()
    xd, dyd = x.to(cuda, torch.bfloat16).contiguous(), dy.to(cuda, torch.bfloat16).contiguous()
    master = w.reshape(k * k, C).to(cuda).contiguous()          # f32 master [k*k][C]
    # ---- weight gradient
    p = _C.DwProblem()
    p.k, p.stride, p.pad_top, p.pad_left, p.act, p.num_segments = k, s, pt, pl, 0, 1
    sg = p.seg[0]
    sg.x, sg.y = xd.data_ptr(), dyd.data_ptr()
    sg.N, sg.H, sg.W, sg.C, sg.Ho, sg.Wo = N, H, W, C, Ho, Wo
    ws = torch.empty(max(lib.rn_depthwise_wgrad_workspace_bytes(ctypes.byref(p)), 16), dtype=torch.uint8, device=cuda)
    dw = torch.empty((k * k, C), dtype=torch.float32, device=cuda)
    _C.check(lib.rn_depthwise_conv2d_nhwc_wgrad(ctypes.byref(p), _C.ptr(dw), _C.ptr(ws), ws.numel(), st), "dw wgrad")
    torch.cuda.synchronize()
    want_dw = wa.grad.reshape(k * k, C).float()
    torch.testing.assert_close(dw.cpu(), want_dw, rtol=2e-3, atol=2e-3 * want_dw.abs().max().item())
    # ---- data gradient
    wflip = torch.empty((k * k, C), dtype=torch.bfloat16, device=cuda)
    _C.check(lib.rn_pack_depthwise_weight_flip(_C.ptr(master), k, C, _C.ptr(wflip), st), "flip")
    src = dyd
    if s == 2:
        src = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=cuda)
        _C.check(lib.rn_upsample_zero2x(_C.ptr(dyd), _C.ptr(src), N, Ho, Wo, C, H, W, st), "up")
    old = _bf(torch.randn((N, H, W, C), generator=g)).to(cuda, torch.bfloat16)   # an already written gradient
    dx = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=cuda)
    q = _C.DwProblem()
    q.k, q.stride, q.pad_top, q.pad_left, q.act, q.num_segments = k, 1, k - 1 - pt, k - 1 - pl, 0, 1
    sq = q.seg[0]
    sq.x, sq.w, sq.y, sq.residual = src.data_ptr(), wflip.data_ptr(), dx.data_ptr(), old.data_ptr()
    sq.N, sq.H, sq.W, sq.C, sq.Ho, sq.Wo = N, H, W, C, H, W
    _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(ctypes.byref(q), st), "dw dgrad")
    torch.cuda.synchronize()
    want_dx = (xa.grad + old.float().cpu().double()).float()
    got = dx.float().cpu()
    assert ((got - want_dx).abs() <= 2.0 ** -7 * want_dx.abs().clamp_min(0.5)).all(), (got - want_dx).abs().max().item()


@pytest.mark.parametrize("N,HW,C,se", [(4, 20 * 20, 144, 6), (3, 10 * 10, 816, 34), (2, 49, 96, 4)])
def test_squeeze_excite_train_forward_backward(cuda, N, HW, C, se):
    from retinanet import _C
    lib = _C.lib()
    g = torch.Generator().manual_seed(C + 1)
    x = _bf(torch.randn((N, HW, C), generator=g) + 0.3)
    w1 = _bf(torch.randn((se, C), generator=g) * (2.0 / C) ** 0.5)
    b1 = torch.randn(se, generator=g) * 0.1
    w2 = _bf(torch.randn((C, se), generator=g) * (2.0 / se) ** 0.5)
    b2 = torch.randn(C, generator=g) * 0.1
    dy = _bf(torch.randn((N, HW, C), generator=g))
    leaves = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    xa, w1a, b1a, w2a, b2a = leaves
    h = xa.mean(dim=1) @ w1a.t() + b1a
    h = h * torch.sigmoid(h)
    gate = torch.sigmoid(h @ w2a.t() + b2a)
    ya = xa * gate[:, None, :]
    (ya * dy.double()).sum().backward()
    st = _C.current_stream()
    dev = lambda t, dt=None: t.to(cuda, dt).contiguous() if dt else t.to(cuda).contiguous()
    xd, dyd = dev(x, torch.bfloat16), dev(dy, torch.bfloat16)
    w1d, w2d, b1d, b2d = dev(w1, torch.bfloat16), dev(w2, torch.bfloat16), dev(b1), dev(b2)
    nbytes = lib.rn_se_workspace_bytes(N, C)
    state = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    yd = torch.empty_like(xd)
    _C.check(lib.rn_squeeze_excite_fwd(_C.ptr(xd), _C.ptr(yd), N, HW, C, _C.ptr(w1d), _C.ptr(b1d), _C.ptr(w2d),
                                       _C.ptr(b2d), se, _C.ptr(state), nbytes, st), "se fwd")
    dx = torch.empty_like(xd)
    dw1 = torch.empty((se, C), device=cuda); db1 = torch.empty((se,), device=cuda)
    dw2 = torch.empty((C, se), device=cuda); db2 = torch.empty((C,), device=cuda)
    _C.check(lib.rn_squeeze_excite_bwd(_C.ptr(xd), _C.ptr(dyd), _C.ptr(dx), N, HW, C, _C.ptr(w1d), _C.ptr(w2d), se,
                                       _C.ptr(state), _C.ptr(dw1), _C.ptr(db1), _C.ptr(dw2), _C.ptr(db2), _C.ptr(ws),
                                       nbytes, st), "se bwd")
    torch.cuda.synchronize()
    torch.testing.assert_close(yd.float().cpu(), ya.detach().float(), rtol=2 ** -6, atol=1e-2)
    # the kernel differentiates through its bf16-rounded intermediates: compare direction and size
    for name, got, want in (("dx", dx.float().cpu(), xa.grad), ("dw1", dw1.cpu(), w1a.grad), ("db1", db1.cpu(), b1a.grad),
                            ("dw2", dw2.cpu(), w2a.grad), ("db2", db2.cpu(), b2a.grad)):
        a, b = got.double().reshape(-1), want.reshape(-1)
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.999, (name, cos)
        assert abs(float(a.norm() / (b.norm() + 1e-30)) - 1.0) < 0.02, (name, float(a.norm() / b.norm()))


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


# ---- training (a18: MBConv / SE / separable convs backward) --------------------------------------------
def _engine_grad(eng, k):
    """gradient of variable k in the Keras layout"""
    got = eng._pview(k, eng.G)
    kind = eng.var_kind.get(k, ("other", None))[0]
    shape = eng.model.variables[k].shape
    if kind in ("conv", "se1", "se2"):
        kh, kw, ci, co = shape
        got = got.reshape(co, kh, kw, ci).permute(1, 2, 3, 0)
    return got.reshape(shape).cpu()


def _zero_gradient_betas(name):
    """beta of a block's project BatchNorm when the block's output is read only through convs that are
    batch-normalised again (the next block's expand conv; the FPN's lateral / P6 convs): a per-channel constant
    in front of conv + training-mode BatchNorm has analytically zero gradient.  It carries signal only when the NEXT
    block adds it through its skip connection."""
    from retinanet.model.backbone.efficientnet import block_table
    blocks = block_table(name)
    out = set()
    for i, b in enumerate(blocks):
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        next_skips = nxt is not None and nxt["stride"] == 1 and nxt["cin"] == nxt["cout"]
        if not next_skips:
            last_bn = "tpu_batch_normalization_1" if b["expand"] == 1 else "tpu_batch_normalization_2"
            out.add(f"{name}/blocks_{i}/{last_bn}/beta")
    return out


def run_wiring(cuda, name, size, B, upstream, with_floor=False):
    """-> (forward relative errors per output, gradient rows (cos, norm ratio, ref norm, name), losses or None);
    with_floor: also the same two for the restatement evaluated in float32 against its float64 evaluation — the
    arithmetic noise floor of the comparison (tools/oracle_noise_floor.py)."""
    from make_golden import synth_gt
    from model_ref import RefTrainer
    from retinanet.cfg import efficientnet_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    p = efficientnet_params(name, input_size=size)
    p.architecture.batch_norm.use_sync = False
    builder = ModelBuilder(p, "train", device=cuda, seed=5)
    model = builder()
    g = torch.Generator().manual_seed(5)
    for k, v in model.variables.items():
        if k.endswith("/gamma"):
            # the last BN of every MBConv block gets a small gamma: keeps the 16-block residual chain
            # well conditioned under bf16 noise (same reasoning as the ResNet-26 wiring test)
            last = k.endswith("tpu_batch_normalization_2/gamma") or k.endswith("blocks_0/tpu_batch_normalization_1/gamma")
            lo, span = (0.1, 0.2) if last else (0.75, 0.5)
            v.copy_((torch.rand(v.shape, generator=g) * span + lo).to(cuda))
        elif k.endswith("/beta"):
            v.copy_((torch.randn(v.shape, generator=g) * 0.1).to(cuda))
    eng = TrainEngine(model, B, frozen_regexes=[])
    ref = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True)
    ref32 = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True, dtype=torch.float32) if with_floor else None
    images = torch.randn((B, size, size, 3), generator=g)
    # drop_connect: fix the per-image factors (one block dropped for image 0, one for image 1) and hand the
    # same factors to the restatement.  Skip blocks: 9 of EfficientNet-B0's 16, 19 of B3's 26 (efficientnet.py:824-827)
    assert len(eng.dc_masks) == {"efficientnet-b0": 9, "efficientnet-b3": 19}[name]
    assert all(0.8 <= sp < 1.0 for _, sp in eng.dc_masks.values())
    for j, (out, (m, sp)) in enumerate(sorted(eng.dc_masks.items(), key=lambda kv: int(kv[0][1:].split("_")[0]))):
        m.copy_(torch.tensor([0.0 if (j % 4 == b) else 1.0 / sp for b in range(B)]))
    ref.drop_connect_factors = {int(out[1:].split("_")[0]): m.cpu().double() for out, (m, sp) in eng.dc_masks.items()}
    if ref32 is not None:
        ref32.drop_connect_factors = ref.drop_connect_factors
    preds = eng.forward(images.to(cuda), draw=False)
    losses = None
    if upstream == "dense":
        up = {k: {lv: torch.randn(preds[k][lv].shape, generator=g) for lv in preds[k]} for k in preds}
        eng.backward({k: {lv: t.to(cuda) for lv, t in d.items()} for k, d in up.items()})
    else:
        enc = LabelEncoder(p, device=cuda)
        rng = np.random.default_rng(5)
        gts = [synth_gt(rng, int(rng.integers(2, 9)), size) for _ in range(B)]
        Gmax = max(x[0].shape[0] for x in gts)
        gb, gc, cnt = np.zeros([B, Gmax, 4], np.float32), np.zeros([B, Gmax], np.float32), np.zeros([B], np.int32)
        for i, (b, c) in enumerate(gts):
            gb[i, :len(b)], gc[i, :len(c)], cnt[i] = b, c, len(b)
        targets = enc.encode_batch(torch.from_numpy(gb), torch.from_numpy(gc), torch.from_numpy(cnt))
        loss = model.loss(targets, preds, compute_grads=True, grad_scale=1.0)
        eng.backward(model.loss.grads)
    torch.cuda.synchronize()
    rp = ref.forward_train(images)
    fwd = {}
    for k in rp:
        for lv in rp[k]:
            a, b = preds[k][lv].float().cpu().double().reshape(-1), rp[k][lv].detach().reshape(-1)
            fwd[(k, lv)] = ((a - b).norm() / (b.norm() + 1e-30)).item()
    if upstream == "dense":
        sum((rp[k][lv] * up[k][lv].double()).sum() for k in up for lv in up[k]).backward()
    else:
        rl = ref.loss(rp, targets["_flat"]["class-targets"].cpu().numpy(), targets["_flat"]["box-targets"].cpu().numpy(),
                      float(targets["num-positives"].sum().item()))
        losses = {k: (loss[k].item(), float(rl[k].detach())) for k in ("box-loss", "class-loss", "weighted-loss")}
        rl["weighted-loss"].backward()
    assert set(eng.train_names) == set(ref.leaf)

    def table(grad_of):
        rows = []
        for k in eng.train_names:
            want = ref.leaf[k].grad
            a, b = grad_of(k).double().reshape(-1), want.reshape(-1)
            rows.append((float(a @ b / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30)),
                         float(b.norm()), k))
        return rows
    rows = table(lambda k: _engine_grad(eng, k))
    if not with_floor:
        return fwd, rows, losses
    rp32 = ref32.forward_train(images)
    fwd32 = {(k, lv): ((rp32[k][lv].detach().double().reshape(-1) - rp[k][lv].detach().reshape(-1)).norm()
                       / (rp[k][lv].detach().norm() + 1e-30)).item() for k in rp for lv in rp[k]}
    if upstream == "dense":
        sum((rp32[k][lv] * up[k][lv]).sum() for k in up for lv in up[k]).backward()
    else:
        ref32.loss(rp32, targets["_flat"]["class-targets"].cpu().numpy(), targets["_flat"]["box-targets"].cpu().numpy(),
                   float(targets["num-positives"].sum().item()))["weighted-loss"].backward()
    return fwd, rows, losses, fwd32, table(lambda k: ref32.leaf[k].grad)


def _signal_rows(rows, name):
    """tensors that carry signal: not a bias in front of a BatchNorm, not an analytically-zero beta, and at least
    0.3 x the median gradient norm"""
    zero = _zero_gradient_betas(name)
    rows = [r for r in rows if r[3] not in zero and not (r[3].endswith("/bias") and "prediction" not in r[3] and "/se/" not in r[3])]
    med = float(np.median([r[2] for r in rows]))
    return {r[3]: r for r in rows if r[2] >= 0.3 * med}


def test_efficientnet_backward_wiring(cuda):
    """Whole-network backward (separable heads -> separable FPN -> MBConv backbone incl. the 3x3 stem) for a dense
    random upstream gradient, against autograd through the bf16-emulating CPU restatement in float64, with fixed
    drop_connect factors (some blocks dropped per image).  EfficientNet-B0 256^2, batch 2: 49 convs and 49
    training-mode BatchNorms amplify every flipped bf16 rounding, so the tolerance is the restatement's own
    arithmetic noise — its float32 evaluation against its float64 evaluation with the same rounding points
    (tools/oracle_noise_floor.py: forward 13 %, gradient cosine median 0.88; the HIP path measures 10 %, 0.91)."""
    name = "efficientnet-b0"
    fwd, rows, _, fwd_floor, rows_floor = run_wiring(cuda, name, 256, 2, "dense", with_floor=True)
    assert max(fwd.values()) <= 1.3 * max(fwd_floor.values()) + 0.005, (max(fwd.values()), max(fwd_floor.values()))
    sig, sig_floor = _signal_rows(rows, name), _signal_rows(rows_floor, name)
    assert len(sig) > 0.5 * len(rows)
    cos = np.array([r[0] for r in sig.values()])
    cos_floor = np.array([sig_floor[k][0] for k in sig if k in sig_floor])
    assert np.median(cos) > np.median(cos_floor) - 0.02, (np.median(cos), np.median(cos_floor))
    assert np.quantile(cos, 0.05) > np.quantile(cos_floor, 0.05) - 0.10, (np.quantile(cos, 0.05), np.quantile(cos_floor, 0.05))
    ratios = np.array([r[1] for r in sig.values()])
    assert np.median(np.abs(ratios - 1)) < 0.08, np.median(np.abs(ratios - 1))
    by = {r[3]: r[0] for r in rows}
    assert by["class-head/class-head-prediction-conv2d/pointwise_kernel"] > 0.995
    assert by["box-head/box-head-prediction-conv2d/depthwise_kernel"] > 0.99
    assert by[name + "/stem/conv2d/kernel"] > 0.85


def test_config4_efficientnet_b3_640_train_step(cuda):
    """BASELINE configs[4] at full size — EfficientNet-B3, 640 x 640, separable FPN / heads, a shard of 2 images (the
    float64 restatement of more takes minutes) — one forward + RetinaNetLoss + backward with the real targets against
    the float64 restatement.  Same criterion as the small case: the restatement's own float32 evaluation is the
    noise floor (26 MBConv blocks: forward 23 %, gradient-cosine median 0.79 — tools/oracle_noise_floor.py
    efficientnet-b3 640 2 loss); the HIP path may be 1.3 x as far on the forward outputs and must keep the gradient
    direction the floor keeps."""
    name = "efficientnet-b3"
    fwd, rows, losses, fwd_floor, rows_floor = run_wiring(cuda, name, 640, 2, "loss", with_floor=True)
    for k, (got, want) in losses.items():
        assert got == pytest.approx(want, rel=0.01), k
    assert max(fwd.values()) <= 1.3 * max(fwd_floor.values()) + 0.005, (max(fwd.values()), max(fwd_floor.values()))
    sig, sig_floor = _signal_rows(rows, name), _signal_rows(rows_floor, name)
    cos = np.array([r[0] for r in sig.values()])
    cos_floor = np.array([sig_floor[k][0] for k in sig if k in sig_floor])
    print("gradient cosine median: HIP %.4f, float32 restatement %.4f" % (np.median(cos), np.median(cos_floor)))
    assert np.median(cos) > np.median(cos_floor) - 0.05, (np.median(cos), np.median(cos_floor))
    by, by_floor = {r[3]: r[0] for r in rows}, {r[3]: r[0] for r in rows_floor}
    assert by["class-head/class-head-prediction-conv2d/pointwise_kernel"] > 0.995
    assert by["class-head/class-head-prediction-conv2d/bias"] > 0.999
    # the box prediction layer sees a sparse gradient (a handful of positive anchors in 2 images): its cosine moves
    # with the noise of the 4-layer tower below it — bound it by the float32 restatement's own value
    k = "box-head/box-head-prediction-conv2d/pointwise_kernel"
    assert by[k] > min(0.99, by_floor[k] - 0.02), (by[k], by_floor[k])


def test_efficientnet_train_steps_reduce_loss(cuda):
    """Full training steps (encode -> forward -> loss -> backward -> clip -> SGD/EMA) on a fixed batch:
    the loss must fall, every parameter tensor must move, and the compute copies must track the masters."""
    from make_golden import synth_gt
    from retinanet.cfg import efficientnet_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    size, B = 256, 4
    p = efficientnet_params("efficientnet-b0", input_size=size)
    p.architecture.batch_norm.use_sync = False
    builder = ModelBuilder(p, "train", device=cuda, seed=11)
    model = builder()
    eng = TrainEngine(model, B, frozen_regexes=[])
    enc = LabelEncoder(p, device=cuda)
    rng = np.random.default_rng(11)
    gts = [synth_gt(rng, int(rng.integers(2, 9)), size) for _ in range(B)]
    Gmax = max(x[0].shape[0] for x in gts)
    gb, gc, cnt = np.zeros([B, Gmax, 4], np.float32), np.zeros([B, Gmax], np.float32), np.zeros([B], np.int32)
    for i, (b, c) in enumerate(gts):
        gb[i, :len(b)], gc[i, :len(c)], cnt[i] = b, c, len(b)
    targets = enc.encode_batch(torch.from_numpy(gb), torch.from_numpy(gc), torch.from_numpy(cnt))
    images = torch.randn((B, size, size, 3), generator=torch.Generator().manual_seed(11)).to(cuda)
    w0 = eng.P.clone()
    losses = []
    model.optimizer.lr = lambda step: 0.01      # the schedule's warm-up start is too small to see in 12 steps
    for _ in range(12):
        out = eng.train_step(images, targets)
        losses.append(float(out["weighted-loss"].item()))
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    # stochastic depth is active (random per-image block drops): the fall is noisier than without it
    assert len(eng.dc_masks) == 9
    assert np.mean(losses[-3:]) < 0.97 * np.mean(losses[:3]), losses
    moved = [(k, float((eng._pview(k) - w0[eng.p_off[k][0]:eng.p_off[k][0] + eng.p_off[k][1]]).abs().max()))
             for k in eng.train_names]
    # every kernel must move (BatchNorm parameters of the 1x1 / 2x2 pyramid levels see gradients below the
    # fp32 resolution of gamma = 1 at this learning rate)
    still = [k for k, d in moved if d == 0.0 and "kernel" in k.rsplit("/", 1)[-1]]
    assert not still, still[:10]
    for (off, n), bfo in eng._bf_copies:   # bf16 compute copies follow the f32 masters
        torch.testing.assert_close(eng.Pbf[bfo:bfo + n].float(), eng.P[off:off + n].to(eng.h16).float(), rtol=0, atol=0)



def test_config4_bench_batch_backward_is_stream_and_run_independent(cuda):
    """BASELINE configs[4] at the bench's own shard — EfficientNet-B3, 640 x 640, 32 images, mixed_float16 — forward and
    backward through the default paths (fused squeeze-excite steps, slab plans of the BatchNorm / pooling reductions picked
    from this batch's chunk counts, depthwise weight gradients, two streams): the gradient buffer must be finite,
    populated, identical from run to run and identical to the one-stream order bit for bit.  (The float64 comparison
    above runs 2 images; the launch geometries of 32 exist at no smaller size.)"""
    from retinanet.cfg import efficientnet_params
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    size, B = 640, 32
    p = efficientnet_params("efficientnet-b3", input_size=size)
    p.architecture.batch_norm.use_sync = False
    model = ModelBuilder(p, "train", device=cuda, seed=3)()
    eng = TrainEngine(model, B, frozen_regexes=[])
    assert eng.f16 and eng.side_stream_on
    images = torch.randn((B, size, size, 3), generator=torch.Generator().manual_seed(3)).to(cuda)
    preds = eng.forward(images)
    g = torch.Generator().manual_seed(4)
    up = {k: {lv: (torch.randn(preds[k][lv].shape, generator=g) * 1e-3).to(cuda) for lv in preds[k]} for k in preds}
    grads = []
    try:
        for two_streams in (True, False, True):
            eng.side_stream_on = two_streams
            eng.G.zero_()
            eng.forward(images, draw=False)      # the same stochastic-depth masks every time
            eng.backward(up)
            torch.cuda.synchronize()
            grads.append(eng.G.clone())
    finally:
        eng.side_stream_on = True
    assert torch.isfinite(grads[0]).all()
    assert int((grads[0] != 0).sum()) > grads[0].numel() // 2
    assert torch.equal(grads[0], grads[1]), "two-stream gradients differ from the one-stream order"
    assert torch.equal(grads[0], grads[2]), "two runs of the two-stream order differ"
