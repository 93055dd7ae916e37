"""GPU parity of rn_bottleneck64_fwd (csrc/rn_bneck.hip): one launch for a whole ResNet stage-1 bottleneck block in
inference form — retinanet/model/backbone/resnet.py:194-248 with filters = 64 — against
  (1) a float64 restatement with the rounding points of the per-layer path (include/rnet_hip.h rn_conv_segment: Conv2D output
      -> bf16, BatchNorm -> bf16, + shortcut, relu -> bf16), and
  (2) the per-layer HIP path itself: three (four) rn_conv2d_nhwc_fwd launches on the same weights.
The fused kernel sums its K dimension in another order than the per-layer kernels, so an output may sit one 16-bit step off
where a sum straddles a rounding boundary — in a, b or the output itself: all but a small fraction of the elements within one
step, every element within four."""
import ctypes
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import _conv_gpu

pytestmark = pytest.mark.gpu
_DT = {"bf16": torch.bfloat16, "f16": torch.float16}


def _block(seed, N, H, W, Cx, dt):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = F.relu(rnd(N, H, W, Cx)) * 1.5                    # a post-relu activation, like the block's real input
    w = {"a": rnd(1, 1, Cx, 64) / Cx ** 0.5 * 1.4, "b": rnd(3, 3, 64, 64) / (9 * 64) ** 0.5 * 1.4,
         "o": rnd(1, 1, 64, 256) / 8.0 * 1.2}
    if Cx == 64:
        w["s"] = rnd(1, 1, 64, 256) / 8.0
    aff = {}
    for k, C in (("a", 64), ("b", 64), ("o", 256), ("s", 256)):
        aff[k] = (torch.rand(C, generator=g) * 0.8 + 0.6, rnd(C) * 0.3)
    return x.to(dt), w, aff


def _ref(x, w, aff, dt):
    rb = lambda t: t.float().to(dt).double()
    cw = lambda t: rb(t).permute(3, 2, 0, 1)               # HWIO -> OIHW, rounded like the packed weights
    bn = lambda t, k: rb(rb(t) * aff[k][0].float().double().view(1, -1, 1, 1) + aff[k][1].float().double().view(1, -1, 1, 1))
    xx = x.double().permute(0, 3, 1, 2)
    a = F.relu(bn(F.conv2d(xx, cw(w["a"])), "a"))
    b = F.relu(bn(F.conv2d(a, cw(w["b"]), padding=1), "b"))
    o = bn(F.conv2d(b, cw(w["o"])), "o")
    sc = bn(F.conv2d(xx, cw(w["s"])), "s") if "s" in w else xx
    return rb(F.relu(o + sc)).permute(0, 2, 3, 1)


def _fused(cuda, lib, x, w, aff, dt, opts=None):
    from retinanet import _C
    N, H, W, Cx = x.shape
    assert lib.rn_bottleneck64_supported(N, H, W, Cx) == 1
    dev = lambda t: t.to(cuda).float().contiguous()
    wa, wb, wo = dev(w["a"]), dev(w["b"]), dev(w["o"])
    ws = dev(w["s"]) if "s" in w else None
    packed = torch.empty((lib.rn_bottleneck64_packed_bytes(Cx),), dtype=torch.uint8, device=cuda)
    _C.check(lib.rn_bottleneck64_pack(_C.ptr(wa), _C.ptr(wb), _C.ptr(wo), _C.ptr(ws), Cx, _C.ptr(packed), _C.current_stream()))
    parts = [aff["a"][0], aff["a"][1], aff["b"][0], aff["b"][1], aff["o"][0], aff["o"][1]]
    if Cx == 64:
        parts += [aff["s"][0], aff["s"][1]]
    affine = torch.cat([p.float() for p in parts]).to(cuda).contiguous()
    xd = x.to(cuda).contiguous()
    y = torch.full((N, H, W, 256), float("nan"), dtype=dt, device=cuda)
    p = _C.Bottleneck64Problem()
    p.x, p.y, p.w_packed, p.affine = xd.data_ptr(), y.data_ptr(), packed.data_ptr(), affine.data_ptr()
    p.N, p.H, p.W, p.Cx = N, H, W, Cx
    if opts:
        p.opts = _C.LaunchOpts(**opts)
    _C.check(lib.rn_bottleneck64_fwd(ctypes.byref(p), _C.current_stream()), "rn_bottleneck64_fwd")
    torch.cuda.synchronize()
    return y.float().cpu()


def _steps_off(got, want, dt):
    """|got - want| in units of the 16-bit type's spacing at max(|want|, the tensor's rms): the output is relu(o + shortcut)
    with both addends bf16 tensors of that magnitude, so an output near zero carries the rounding of its addends, not of
    its own value"""
    mant = 7 if dt == torch.bfloat16 else 10
    w = want.double().abs()
    mag = torch.maximum(w, want.double().pow(2).mean().sqrt())
    ulp = torch.pow(2.0, torch.floor(torch.log2(mag)) - mant)
    return (got.double() - want.double()).abs() / ulp


CASES = [
    # (N, H, W, Cx, launch opts)
    (2, 7, 64, 256, None), (1, 5, 32, 64, None), (3, 9, 96, 256, None), (2, 6, 128, 64, None),
    (1, 40, 160, 256, None), (2, 23, 160, 64, None),            # W = 160: three ring rows, two barriers per row
    (1, 11, 128, 256, None), (1, 4, 128, 64, None),             # W <= 128: four ring rows, one barrier per row
    (1, 1, 32, 256, None), (1, 2, 64, 64, None),                # fewer rows than the halo: every stage-A row is a border case
    (5, 33, 64, 256, {"max_workgroups": 7}),                    # few workgroups: long row segments, uneven last segment
    (2, 160, 160, 256, None), (1, 160, 160, 64, None),          # BASELINE configs[0] / [1] geometry (640 x 640 input)
]


@pytest.mark.parametrize("build", ["bf16", "f16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:4]))
def test_bottleneck64_vs_float64_and_per_layer_path(cuda, build, case):
    from retinanet import _C
    N, H, W, Cx, opts = case
    if build == "f16" and H * W > 64 * 64:
        pytest.skip("the half build runs the small cases")
    dt = _DT[build]
    lib = _C.lib(build == "f16")
    x, w, aff = _block(zlib.crc32(repr(case).encode()) % 2 ** 31, N, H, W, Cx, dt)
    got = _fused(cuda, lib, x, w, aff, dt, opts)
    assert torch.isfinite(got).all()
    want = _ref(x, w, aff, dt)
    off = _steps_off(got, want, dt)
    assert (off > 1.0).double().mean().item() < 4e-3 and off.max().item() <= 4.0, (off.max().item(), (off > 1).double().mean().item())
    assert (got > 0).double().mean().item() > 0.2          # the block is alive
    if build != "bf16":
        return
    # the per-layer path on the bfloat16 build: the launches the fused one replaces
    seg = lambda xin, k, wk, key, res=None: {"x": xin, "w": w[wk], "scale": aff[key][0], "shift": aff[key][1], "residual": res}
    a = _conv_gpu(cuda, [seg(x.float(), 1, "a", "a")], 1, 1, 0, "relu", False)[0]
    b = _conv_gpu(cuda, [seg(a, 3, "b", "b")], 3, 1, 1, "relu", False)[0]
    sc = _conv_gpu(cuda, [seg(x.float(), 1, "s", "s")], 1, 1, 0, None, False)[0] if Cx == 64 else x.float()
    per_layer = _conv_gpu(cuda, [seg(b, 1, "o", "o", res=sc)], 1, 1, 0, "relu", False)[0]
    off2 = _steps_off(got, per_layer, dt)
    assert (off2 > 1.0).double().mean().item() < 4e-3 and off2.max().item() <= 4.0, (off2.max().item(),)


def test_bottleneck64_is_deterministic_and_refuses_other_shapes(cuda):
    from retinanet import _C
    lib = _C.lib()
    x, w, aff = _block(3, 2, 12, 160, 256, torch.bfloat16)
    runs = [_fused(cuda, lib, x, w, aff, torch.bfloat16) for _ in range(3)]
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    # a different cut of the rows into workgroups computes the same bits (every row's sums are its own)
    other = _fused(cuda, lib, x, w, aff, torch.bfloat16, {"max_workgroups": 3})
    assert torch.equal(runs[0], other)
    for bad in ((1, 8, 48, 256), (1, 8, 256, 256), (1, 8, 192, 256), (1, 8, 64, 128), (0, 8, 64, 256)):
        assert lib.rn_bottleneck64_supported(*bad) == 0
    p = _C.Bottleneck64Problem()
    p.N, p.H, p.W, p.Cx = 1, 8, 48, 256
    assert lib.rn_bottleneck64_fwd(ctypes.byref(p), _C.current_stream()) == _C.RN_EINVAL


def test_engines_use_the_fused_blocks(cuda, monkeypatch):
    """InferenceEngine: the three stage-1 blocks of ResNet-50 are three launches (ten before), the feature map they produce
    agrees with the per-layer engine's to a 16-bit step; TrainEngine: fused exactly when the blocks are frozen
    (`resnet_initial`), never when they train"""
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    from retinanet.model.engine import InferenceEngine
    from retinanet.model.train_engine import TrainEngine
    p = default_params(input_size=256)
    b = ModelBuilder(p, "val", device=cuda)
    model = b()
    g = torch.Generator().manual_seed(4)
    for k, v in model.variables.items():
        if k.endswith("/gamma"):
            v.copy_((torch.rand(v.shape, generator=g) * 0.5 + 0.75).to(cuda))
        elif k.endswith("/beta") or k.endswith("/moving_mean"):
            v.copy_((torch.randn(v.shape, generator=g) * 0.1).to(cuda))
        elif k.endswith("/moving_variance"):
            v.copy_((torch.rand(v.shape, generator=g) * 0.5 + 0.75).to(cuda))
    images = torch.randn((2, 256, 256, 3), generator=g).to(cuda)
    eps = p.architecture.batch_norm.epsilon
    fused = InferenceEngine(model.graph, model.variables, 2, cuda, bn_epsilon=eps)
    monkeypatch.setenv("RNET_FUSE_BOTTLENECK", "0")
    plain = InferenceEngine(model.graph, model.variables, 2, cuda, bn_epsilon=eps)
    monkeypatch.delenv("RNET_FUSE_BOTTLENECK")
    names_f, names_p = [n for _, n in fused.steps], [n for _, n in plain.steps]
    assert [n for n in names_f if n.startswith("bneck:")] == ["bneck:g1b0_out", "bneck:g1b1_out", "bneck:g1b2_out"]
    assert len(names_p) - len(names_f) == 7 and not [n for n in names_p if n.startswith("bneck:")]
    of, op_ = fused(images), plain(images)
    torch.cuda.synchronize()
    off = _steps_off(fused.t["g1b2_out"].float().cpu(), plain.t["g1b2_out"].float().cpu(), torch.bfloat16)
    assert (off > 1.0).double().mean().item() < 1e-2 and off.max().item() <= 8.0, (off.max().item(),)
    for key in of:
        for lv in of[key]:
            a_, b_ = of[key][lv].float(), op_[key][lv].float()
            assert (a_ - b_).abs().max().item() <= 0.02 * b_.abs().max().item() + 1e-3, (key, lv)
    # training: frozen stage 1 -> fused; everything trainable -> per layer
    pt = default_params(input_size=256)
    pt.architecture.batch_norm.use_sync = False
    bt = ModelBuilder(pt, "train", device=cuda)
    mt = bt()
    for k, v in mt.variables.items():
        v.copy_(model.variables[k])
    rx = [bt.FREEZE_VARS_REGEX[n] for n in pt.training.freeze_variables]
    assert rx, "the default configuration freezes resnet_initial"
    eng = TrainEngine(mt, 2, frozen_regexes=rx)
    assert sorted(fb.name for fb in eng.bneck.values()) == ["bneck:g1b0_out", "bneck:g1b1_out", "bneck:g1b2_out"]
    eng_live = TrainEngine(mt, 2, frozen_regexes=[])
    assert not eng_live.bneck
    # the frozen layers compute the same function in the training forward pass (same weights, inference form)
    eng.forward(images)
    torch.cuda.synchronize()
    off = _steps_off(eng.t["g1b2_out"].float().cpu(), plain.t["g1b2_out"].float().cpu(), torch.bfloat16)
    assert (off > 1.0).double().mean().item() < 1e-2 and off.max().item() <= 8.0, (off.max().item(),)
