"""GPU parity of one whole training step (a10 + backward of a5-a9) against the float64 CPU
autograd restatement oracle/model_ref.py::RefTrainer."""
import numpy as np
import pytest
import torch

from make_golden import synth_gt
from model_ref import RefTrainer

pytestmark = pytest.mark.gpu


def _setup(cuda, size, B, balanced, seed=3, depth=26, freeze=True, precision="mixed_bfloat16", launch_opts=None):
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    p = default_params(input_size=size, balanced=balanced, precision=precision)
    p.architecture.batch_norm.use_sync = False
    # A randomly initialised 50-layer ReLU/BN net with O(1) residual gammas amplifies ANY perturbation
    # (bf16 rounding included) by ~1.3x per block, so element-wise parity with a float64 run is not
    # defined there.  The wiring check uses the bottleneck ResNet-26 variant of the same builder
    # (resnet.py:366-369 _MODEL_CONFIG) with small last-BN gammas: every branch carries signal and
    # gradient, but the map stays well conditioned.
    p.architecture.backbone.depth = depth
    builder = ModelBuilder(p, "train", device=cuda, seed=seed)
    model = builder()
    g = torch.Generator().manual_seed(seed)
    for k, v in model.variables.items():
        if k.endswith("/gamma"):
            zero_init = model.graph.bns[k[:-len("/gamma")]]["gamma_zero"]
            lo, span = (0.1, 0.2) if zero_init else (0.75, 0.5)
            v.copy_((torch.rand(v.shape, generator=g) * span + lo).to(cuda))
        elif k.endswith("/beta"):
            v.copy_((torch.randn(v.shape, generator=g) * 0.1).to(cuda))
        elif "head" in k and k.endswith("/kernel"):
            v.copy_((torch.randn(v.shape, generator=g) * 0.02).to(cuda))
    import re
    rx = [builder.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables]
    if depth == 26:   # stem + block_group1 of ResNet-26 = conv2d .. conv2d_7 (same role as 'resnet_initial')
        rx = [re.compile(r"^(conv2d|batch_normalization)(_[1-7])?/")]
    if not freeze:    # the 30x configs train everything from scratch (training.freeze_variables: [])
        rx = []
    eng = TrainEngine(model, B, frozen_regexes=rx, launch_opts=launch_opts)
    enc = LabelEncoder(p, device=cuda)
    rng = np.random.default_rng(seed)
    gts = [synth_gt(rng, int(rng.integers(2, 9)), size) for _ in range(B)]
    Gmax = max(x[0].shape[0] for x in gts)
    gb, gc, cnt = np.zeros([B, Gmax, 4], np.float32), np.zeros([B, Gmax], np.float32), np.zeros([B], np.int32)
    for i, (b, c) in enumerate(gts):
        gb[i, :len(b)], gc[i, :len(c)], cnt[i] = b, c, len(b)
    targets = enc.encode_batch(torch.from_numpy(gb), torch.from_numpy(gc), torch.from_numpy(cnt))
    images = torch.randn((B, size, size, 3), generator=g)
    return p, model, eng, targets, images


def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def _engine_grad(eng, k):
    got = eng._pview(k, eng.G)
    if k.endswith("/kernel"):
        c = eng.g.convs[k[:-len("/kernel")]]
        got = got.reshape(c["cout"], c["k"], c["k"], c["cin"]).permute(1, 2, 3, 0)
    return got.cpu()


def _gradient_rows(grad_of, ref, names):
    rows = {}
    for k in names:
        if k.endswith("/bias") and "prediction" not in k:
            continue   # bias in front of BatchNorm: analytically zero gradient, pure rounding noise
        want = ref.leaf[k].grad
        got = grad_of(k).reshape(want.shape)
        rows[k] = (_cos(got, want), got.double().norm().item() / (want.double().norm().item() + 1e-30))
    return rows


def _kernel_ids(eng):
    import ctypes
    from retinanet import _C
    return {name: eng.lib.rn_conv_kernel_id(ctypes.byref(p)) for name, p in eng.conv_launches}


# the whole-network wiring check runs on both kernel families: "auto" = what the dispatcher picks at these small
# sizes (the 128-row kernels), "persistent" = every eligible layer forced onto the 256-wide persistent kernels
# (conv_big / conv_halo for forward and data gradients with their fused BatchNorm statistics, the 256-wide weight-gradient
# kernels) — the kernels the full-size training bench runs on.  Per-engine rn_launch_opts, not process state.
_PERSISTENT = dict(conv_tile=2, wgrad_kernel=2)


@pytest.mark.parametrize("size,B,balanced,freeze,precision,kernels", [
    (256, 4, True, True, "mixed_bfloat16", "auto"), (256, 3, False, True, "mixed_bfloat16", "auto"),
    (256, 2, True, False, "mixed_bfloat16", "auto"),
    (256, 4, True, True, "mixed_float16", "auto"),      # the half build (librnet_hip_f16.so) against the restatement rounding to half
    (256, 4, True, True, "mixed_bfloat16", "persistent"),
    (256, 4, True, True, "mixed_float16", "persistent"),
])
def test_backward_wiring_dense_upstream(cuda, size, B, balanced, freeze, precision, kernels):
    """Whole-network backward (heads -> BalanceFeatures -> FPN -> ResNet) for a dense random upstream gradient on the
    predictions, against autograd through the bf16-emulating CPU restatement in float64.

    The tolerance is the restatement's OWN arithmetic noise: the same restatement evaluated in float32 (identical bf16
    rounding points, only the fp summation differs) lands as far from its float64 evaluation as any correct
    implementation can be expected to — a randomly initialised net with training-mode BatchNorm amplifies every
    flipped bf16 rounding (tools/oracle_noise_floor.py: ResNet-26 256^2, forward 3.3 %, gradient cosine median 0.971,
    minimum 0.940; the HIP path measures 3.1 %, 0.971, 0.936 — on the 128-row and on the 256-row kernels alike: every
    kernel rounds to bf16 at the same points, include/rnet_hip.h rn_conv_segment).  The HIP path must sit at that
    floor: forward error <= 1.3 x the floor, every tensor's gradient cosine within 0.06 of its floor value, the median
    within 0.015."""
    p, model, eng, targets, images = _setup(cuda, size, B, balanced, freeze=freeze, precision=precision,
                                            launch_opts=_PERSISTENT if kernels == "persistent" else None)
    assert eng.f16 == (precision == "mixed_float16")
    ids = _kernel_ids(eng)
    if kernels == "persistent":
        assert ids["fwd:tower0"] == 2 and ids["dgrad:tower3"] == 2 and any(v == 1 for v in ids.values()), ids
    else:
        assert ids["fwd:tower0"] == 0, ids
    ref = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True)
    ref32 = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True, dtype=torch.float32)
    if not freeze:
        assert "conv2d/kernel" in eng.train_names and "batch_normalization/gamma" in eng.train_names
    preds = eng.forward(images.to(cuda))
    g = torch.Generator().manual_seed(99)
    up = {k: {lv: torch.randn(preds[k][lv].shape, generator=g) for lv in preds[k]} for k in preds}
    eng.backward({k: {lv: t.to(cuda) for lv, t in d.items()} for k, d in up.items()})
    torch.cuda.synchronize()
    rp, rp32 = ref.forward_train(images), ref32.forward_train(images)
    floor_fwd = max(_rel(rp32[k][lv].detach(), rp[k][lv].detach()) for k in up for lv in up[k])
    got_fwd = max(_rel(preds[k][lv].float().cpu(), rp[k][lv].detach()) for k in up for lv in up[k])
    assert got_fwd <= 1.3 * floor_fwd + 0.005, (got_fwd, floor_fwd)
    sum((rp[k][lv] * up[k][lv].double()).sum() for k in up for lv in up[k]).backward()
    sum((rp32[k][lv] * up[k][lv]).sum() for k in up for lv in up[k]).backward()
    assert set(eng.train_names) == set(ref.leaf)
    rows = _gradient_rows(lambda k: _engine_grad(eng, k), ref, eng.train_names)
    floor = _gradient_rows(lambda k: ref32.leaf[k].grad, ref, eng.train_names)
    worst = sorted((rows[k][0] - floor[k][0], k, rows[k][0], floor[k][0]) for k in rows)
    assert worst[0][0] > -0.06, worst[:5]
    med, med_floor = np.median([r[0] for r in rows.values()]), np.median([r[0] for r in floor.values()])
    assert med > med_floor - 0.015, (med, med_floor)
    ratios = np.array([r[1] for r in rows.values()])
    assert np.median(np.abs(ratios - 1)) < 0.03 and np.abs(ratios - 1).max() < 0.35, (ratios.min(), ratios.max())
    # the layers next to the loss see almost no accumulated rounding noise
    assert rows["class-head/class-head-prediction-conv2d/kernel"][0] > 0.995
    assert rows["box-head/box-head-prediction-conv2d/kernel"][0] > 0.995


@pytest.mark.parametrize("size,B", [(640, 8), (1024, 4)], ids=["config2-resnet50-640-b8", "config3-resnet50-1024-b4"])
def test_train_step_at_baseline_sizes(cuda, size, B):
    """One training step of BASELINE configs[2] / configs[3] at full depth and resolution (ResNet-50, `resnet_initial`
    frozen, BalanceFeatures; a per-GPU shard of 8 / 4 images), no debug overrides: the dispatcher itself puts the
    wide layers on conv_big / conv_halo / wgrad_big.  Against the float32 autograd restatement with the same bf16
    rounding points: losses, the gradients of the layers next to the loss, per-tensor gradient direction, and the
    global gradient norm the clip sees."""
    import ctypes
    from retinanet import _C
    p, model, eng, targets, images = _setup(cuda, size, B, True, depth=50, freeze=True)
    ids = _kernel_ids(eng)
    # (3 = the 512 x 128 tiles of conv_halo_kernel, which the dispatcher prefers wherever the width is a multiple of 128)
    assert ids["fwd:tower0"] == 3 and ids["dgrad:tower3"] == 3 and ids["fwd:pred_class"] == 3, ids
    assert ids["dgrad:pred_class"] == 3 and ids["fwd:fpn_out"] == 3 and ids["fwd:pred_box"] == 3, ids
    # residual 1x1 (128 -> 512) forward and the data gradient of the 1x1 in front of it (dx has 512 channels)
    assert ids["fwd:g2b1_out"] == 1 and ids["dgrad:g2b1_a"] == 1, ids
    assert all(B * s.H * s.W < (1 << 22) for _, pr in eng.conv_launches for s in [pr.seg[0]])   # rn_fdiv's bound
    ref = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True, dtype=torch.float32)
    preds = eng.forward(images.to(cuda))
    loss = model.loss(targets, preds, compute_grads=True, grad_scale=1.0)
    eng.backward(model.loss.grads)
    torch.cuda.synchronize()
    rp = ref.forward_train(images)
    # bounds = the restatement's own float32-vs-float64 noise at this configuration (tools/oracle_noise_floor.py 50
    # <size> <batch> loss, CPU): forward relative error / gradient cosine median / 5 %-quantile
    #   640 x 640 : 0.0554 / 0.929 / 0.900   (the HIP path measures 0.061 / 0.932)
    #   1024 x 1024: 0.0622 / 0.9175 / 0.873
    floor_fwd, floor_med, floor_q05 = {640: (0.0554, 0.929, 0.900), 1024: (0.0622, 0.9175, 0.8726)}[size]
    for k in ("class-predictions", "box-predictions"):
        for lv in rp[k]:
            assert _rel(preds[k][lv].float().cpu(), rp[k][lv].detach()) < 1.3 * floor_fwd, (k, lv)
    rl = ref.loss(rp, targets["_flat"]["class-targets"].cpu().numpy(), targets["_flat"]["box-targets"].cpu().numpy(),
                  float(targets["num-positives"].sum().item()))
    for k in ("box-loss", "class-loss", "weighted-loss"):
        assert loss[k].item() == pytest.approx(float(rl[k].detach()), rel=0.02), k
    rl["weighted-loss"].backward()
    rows = []
    for k in eng.train_names:
        if k.endswith("/bias") and "prediction" not in k:
            continue   # bias in front of BatchNorm: analytically zero gradient
        want = ref.leaf[k].grad
        if float(want.abs().max()) == 0.0:
            continue   # e.g. the 5x5 level's box-head BatchNorm when no anchor of that level is positive
        got = _engine_grad(eng, k).reshape(want.shape)
        rows.append((_cos(got, want), got.double().norm().item() / (want.double().norm().item() + 1e-30), k))
    rows.sort()
    by = {r[2]: r for r in rows}
    for k in ("class-head/class-head-prediction-conv2d/kernel", "box-head/box-head-prediction-conv2d/kernel",
              "class-head/class-head-prediction-conv2d/bias", "box-head/box-head-prediction-conv2d/bias"):
        assert by[k][0] > 0.995 and abs(by[k][1] - 1) < 0.02, by[k]
    print("gradient cosine: min %.4f (%s), median %.4f" % (rows[0][0], rows[0][2], np.median([r[0] for r in rows])))
    assert rows[len(rows) // 20][0] > floor_q05 - 0.05, rows[:len(rows) // 20 + 1]
    assert np.median([r[0] for r in rows]) > floor_med - 0.025, np.median([r[0] for r in rows])
    gn_got = float(torch.sqrt(sum((_engine_grad(eng, k).double() ** 2).sum() for k in eng.train_names)))
    gn_want = float(torch.sqrt(sum((ref.leaf[k].grad.double() ** 2).sum() for k in eng.train_names)))
    assert gn_got == pytest.approx(gn_want, rel=0.03)


def test_two_stream_backward_is_bit_identical(cuda):
    """TrainEngine.backward runs the weight / bias gradient launches on a second HIP stream (RNET_WGRAD_STREAM,
    default on).  Every kernel is deterministic, so the gradient buffer must equal the one-stream order's bit for
    bit — a race on dy, a saved activation or a workspace would show up as a difference."""
    p, model, eng, targets, images = _setup(cuda, 256, 4, True, freeze=True)
    assert eng.side_stream_on
    assert any(getattr(fn, "side", False) for fn in eng.bwd_steps)
    preds = eng.forward(images.to(cuda))
    g = torch.Generator().manual_seed(99)
    up = {k: {lv: torch.randn(preds[k][lv].shape, generator=g).to(cuda) for lv in preds[k]} for k in preds}
    grads = []
    try:
        for two_streams in (False, True, True, False, True):
            eng.side_stream_on = two_streams
            eng.G.zero_()
            eng.forward(images.to(cuda), draw=False)
            eng.backward(up)
            torch.cuda.synchronize()
            grads.append(eng.G.clone())
    finally:
        eng.side_stream_on = True
    assert int((grads[0] != 0).sum()) > grads[0].numel() // 2
    for i in range(1, len(grads)):
        assert torch.equal(grads[0], grads[i]), f"run {i} differs from the one-stream gradients"


@pytest.mark.parametrize("size,B", [(640, 32), (1024, 16)], ids=["config2-640-b32", "config3-1024-b16"])
def test_bench_batch_backward_is_stream_and_run_independent(cuda, size, B):
    """The bench's own shards end to end (BASELINE configs[2]: ResNet-50, 640 x 640, 32 images on the GPU; configs[3]:
    1024 x 1024, 16 images; `resnet_initial` frozen — the whole-step parity tests above run them at 8 / 4 images):
    forward + backward through every default path (fused stage-1
    blocks, grouped weight gradients, BatchNorm reductions in the data-gradient epilogues, two streams); the gradient
    buffer must be finite, populated, identical from run to run and identical to the one-stream order bit for bit —
    the launch geometries of this batch (tile counts, split-K parts, chunk plans) exist at no smaller size."""
    p, model, eng, targets, images = _setup(cuda, size, B, True, depth=50, freeze=True)
    assert eng.side_stream_on and (len(eng.bneck) > 0 or size != 640)   # (W = 256 at 1024 x 1024: per-layer stage 1)
    preds = eng.forward(images.to(cuda))
    g = torch.Generator().manual_seed(7)
    up = {k: {lv: (torch.randn(preds[k][lv].shape, generator=g) * 1e-3).to(cuda) for lv in preds[k]} for k in preds}
    grads = []
    try:
        for two_streams in (True, False, True):
            eng.side_stream_on = two_streams
            eng.G.zero_()
            eng.forward(images.to(cuda), draw=False)
            eng.backward(up)
            torch.cuda.synchronize()
            grads.append(eng.G.clone())
    finally:
        eng.side_stream_on = True
    assert torch.isfinite(grads[0]).all()
    assert int((grads[0] != 0).sum()) > grads[0].numel() // 2
    assert torch.equal(grads[0], grads[1]), "two-stream gradients differ from the one-stream order"
    assert torch.equal(grads[0], grads[2]), "two runs of the two-stream order differ"


@pytest.mark.parametrize("groups", [False, True], ids=["backbone", "backbone+head-towers"])
def test_bn_backward_reduction_in_the_dgrad_epilogue_matches_the_separate_pass(cuda, monkeypatch, groups):
    """RNET_FUSE_BN_BWD (default on): the data-gradient launch that writes dz of a BatchNorm + ReLU layer with one
    consumer also writes stage 1 of that layer's backward reduction.  dz itself is unchanged; (sum g, sum g*xhat)
    come out of a different fp32 association, so every gradient must agree with the
    separate-pass engine to fp32-summation accuracy (far inside one bf16 ulp of what follows)."""
    monkeypatch.setenv("RNET_FUSE_BN_BWD", "2" if groups else "1")
    p, model, eng, targets, images = _setup(cuda, 256, 4, True, freeze=True)
    # ResNet-26 (2 bottleneck blocks per group, group 1 frozen): conv a -> b of the stride-1 blocks and conv b -> c of
    # every block in groups 2-4; RNET_FUSE_BN_BWD=2: + the ten segments (two heads x five levels) of each of the four
    # head-tower depths (depth 3 completes across the two prediction convs' launches)
    assert len(eng.bn_bwd_fused) == (49 if groups else 9), eng.bn_bwd_fused
    monkeypatch.setenv("RNET_FUSE_BN_BWD", "0")
    from retinanet.model.train_engine import TrainEngine
    import re
    plain = TrainEngine(model, 4, frozen_regexes=[re.compile(r"^(conv2d|batch_normalization)(_[1-7])?/")])
    assert plain.bn_bwd_fused == []
    preds = eng.forward(images.to(cuda))
    g = torch.Generator().manual_seed(7)
    up = {k: {lv: torch.randn(preds[k][lv].shape, generator=g).to(cuda) for lv in preds[k]} for k in preds}
    grads = []
    for e in (eng, plain):
        e.G.zero_()
        e.forward(images.to(cuda), draw=False)
        e.backward(up)
        torch.cuda.synchronize()
        grads.append(e.G.clone())
    assert int((grads[0] != 0).sum()) > grads[0].numel() // 2
    # the first fused layer of the backward order (the last block of group 4, or head-tower depth 3 when the head groups
    # are fused: nothing above it is) receives the same dz in both engines: its gamma / beta gradients ARE the two
    # reductions of identical inputs
    top = next(o for o in eng.ops if o.get("out") == eng.bn_bwd_fused[0])
    for sfx in ("/gamma", "/beta"):
        a, b = eng._pview(top["bn"] + sfx, grads[0]).double(), plain._pview(top["bn"] + sfx, grads[1]).double()
        assert (a - b).norm().item() <= 1e-5 * b.norm().item(), (top["bn"] + sfx, (a - b).norm().item() / b.norm().item())
    # below it the ~1e-6 differences of the sums flip single bf16 roundings of dy, which the net amplifies like any
    # other perturbation (see _setup): the gradients stay the same vectors to a few 1e-3
    worst = max(((eng._pview(k, grads[0]).double() - plain._pview(k, grads[1]).double()).norm().item() /
                 (plain._pview(k, grads[1]).double().norm().item() + 1e-30), k) for k in eng.train_names
                if not (k.endswith("/bias") and "prediction" not in k))   # bias in front of BatchNorm: zero gradient, pure noise
    assert worst[0] < 3e-2, worst


def test_train_step_losses_and_optimizer_arithmetic(cuda):
    """One full step with the real loss: loss values against the restatement, then the optimizer
    stages re-derived in float64 from the engine's own raw gradients (executor.py:401-407,
    optimizers/builder.py:45-54): weight decay, per-tensor + global clip, SGD momentum, EMA."""
    p, model, eng, targets, images = _setup(cuda, 256, 4, True)
    ref = RefTrainer(p, model.variables, frozen_names=eng.frozen, emulate_bf16=True)
    opt = model.optimizer
    lr, dec, mom, clip = opt.lr(0), opt.ema_decay(0), opt.momentum, float(opt.clipnorm)
    alpha = p.training.weight_decay_alpha
    w0 = eng.P.clone()
    preds = eng.forward(images.to(cuda))
    loss = model.loss(targets, preds, compute_grads=True, grad_scale=1.0)
    eng.backward(model.loss.grads)
    raw = eng.G.clone()
    eng.optimizer_step(lr, mom, clip, alpha, dec)
    torch.cuda.synchronize()
    rl = ref.loss(ref.forward_train(images), targets["_flat"]["class-targets"].cpu().numpy(),
                  targets["_flat"]["box-targets"].cpu().numpy(), float(targets["num-positives"].sum().item()))
    for k in ("box-loss", "class-loss", "weighted-loss"):
        assert loss[k].item() == pytest.approx(float(rl[k].detach()), rel=0.03), k
    # float64 replay of the optimizer on the engine's raw gradients
    parts, norms = {}, []
    for k in eng.train_names:
        off, n = eng.p_off[k]
        gk = raw[off:off + n].double().cpu()
        if k.endswith("/kernel"):
            gk = gk + alpha * w0[off:off + n].double().cpu()
        gk = gk * (clip / max(gk.norm().item(), clip))
        parts[k] = gk
    gn = float(np.sqrt(sum(v.norm().item() ** 2 for v in parts.values())))
    F = clip / max(gn, clip)
    assert eng.metrics[1].item() == pytest.approx(gn, rel=1e-4)
    assert eng.metrics[0].item() == pytest.approx(gn * F, rel=1e-4)
    for k in eng.train_names[::7]:
        off, n = eng.p_off[k]
        g = parts[k] * F
        v1 = -lr * g
        w1 = w0[off:off + n].double().cpu() + v1
        e1 = w0[off:off + n].double().cpu() * dec + (1 - dec) * w1
        torch.testing.assert_close(eng.V[off:off + n].double().cpu(), v1, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(eng.P[off:off + n].double().cpu(), w1, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(eng.E[off:off + n].double().cpu(), e1, rtol=1e-5, atol=1e-6)
        if k.endswith("/kernel") and k[:-len("/kernel")] in eng.bf_off:   # bf16 compute copy refreshed by the same kernel
            cname = k[:-len("/kernel")]
            bf = eng.Pbf[eng.bf_off[cname]:eng.bf_off[cname] + n].float().cpu()
            torch.testing.assert_close(bf, w1.float().to(torch.bfloat16).float(), rtol=1 / 128, atol=1e-6)
    # the float32 prediction kernels are re-split into bf16 planes after the step: hi + lo = w to 16 mantissa bits
    for cname, buf in eng.split_pack_of.items():
        c = eng.g.convs[cname]
        off, n = eng.p_off[cname + "/kernel"]
        wm = eng.P[off:off + n].reshape(c["cout"], c["k"], c["k"], c["cin"]).double().cpu()
        planes = buf[:c["cout"]].double().cpu().reshape(c["cout"], c["k"], c["k"], -1, buf.shape[-1] // 2)
        torch.testing.assert_close(planes.sum(dim=3)[..., :c["cin"]], wm, rtol=2.0 ** -15, atol=1e-9)
    bn = next(iter(eng.bn_state))   # moving statistics of a live BN layer (momentum 0.99, Bessel-corrected)
    torch.testing.assert_close(eng.bn_state[bn]["mm"].cpu().double(), ref.new_stats[bn + "/moving_mean"], rtol=0.02, atol=2e-3)
    torch.testing.assert_close(eng.bn_state[bn]["mv"].cpu().double(), ref.new_stats[bn + "/moving_variance"], rtol=0.02, atol=2e-3)


def test_loss_decreases_over_steps(cuda):
    """Twenty steps on one fixed batch: the weighted loss must fall substantially (end-to-end
    sanity of forward, backward, clipping, SGD and the bf16 weight refresh)."""
    p, model, eng, targets, images = _setup(cuda, 128, 2, True, seed=7, depth=50)
    p.training.optimizer.lr_params.warmup_learning_rate = 0.02
    p.training.optimizer.lr_params.initial_learning_rate = 0.02
    from retinanet.optimizers import build_optimizer
    model.optimizer = build_optimizer(p.training.optimizer, p.training.train_steps, p.floatx.precision)
    images = images.to(cuda)
    losses = []
    for _ in range(20):
        losses.append(eng.train_step(images, targets)["weighted-loss"].item())
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < 0.6 * losses[0], losses
    eng.store_to_model(use_ema=False)   # back to the Keras-named variables + inference engine
    preds = model(images, training=False)
    assert torch.isfinite(preds["class-predictions"]["3"]).all()


def test_dynamic_loss_scale_is_applied_one_forward_pass_later(cuda):
    """mixed_float16 (optimizers/builder.py:56-64, LossScaleOptimizer(dynamic=True)): a step whose gradients are not
    finite is dropped ON THE DEVICE (weights, momentum untouched); the host learns about it when the next step has
    enqueued its forward pass (train_engine._resolve_loss_scale) or at finish_step(): the scale halves, the step
    counter does not advance, and the following step runs with the halved scale; `growth_steps` good steps double it."""
    p, model, eng, targets, images = _setup(cuda, 128, 2, True, seed=5, precision="mixed_float16")
    images = images.to(cuda)
    opt = model.optimizer
    assert opt.dynamic_loss_scale and eng.f16
    eng.train_step(images, targets)
    eng.finish_step()
    s0 = eng.loss_scale["scale"]
    assert eng.step_count == 1 and not eng.loss_scale["skipped"] and eng.loss_scale["good"] == 1
    # an overflowing step: a loss scale far beyond the half range makes the scaled upstream gradients infinite
    eng.loss_scale["scale"] = 2.0 ** 40
    before = eng.P.clone()
    out = eng.train_step(images, targets)
    assert eng.step_count == 2 and eng._ls_pending            # optimistic until the flag is looked at
    assert torch.isfinite(out["weighted-loss"])               # the forward pass and the loss are not scaled
    assert torch.equal(eng.P, before)                         # ... and the device dropped the update by itself
    eng.finish_step()
    assert eng.loss_scale["skipped"] and eng.loss_scale["scale"] == 2.0 ** 39 and eng.loss_scale["good"] == 0
    assert eng.step_count == 1 and opt.iterations == 1
    # the next step picks the resolved scale up by itself (no finish_step in between): run until the scale fits again
    eng.loss_scale["scale"] = s0 * 2.0 ** 30
    eng.train_step(images, targets)                           # overflows again
    eng.train_step(images, targets)                           # resolves the step before: halved
    assert eng.loss_scale["scale"] == s0 * 2.0 ** 29 and eng.loss_scale["skipped"]
    eng.finish_step()
    eng.loss_scale["scale"], eng.loss_scale["good"] = s0, eng.loss_scale["growth_steps"] - 1
    n = eng.step_count
    eng.train_step(images, targets)
    eng.finish_step()
    assert eng.step_count == n + 1 and eng.loss_scale["scale"] == 2 * s0 and eng.loss_scale["good"] == 0
    assert not torch.equal(eng.P, before)
