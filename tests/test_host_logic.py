"""CPU: host-side logic that needs no GPU — config attr-dict, graph/variable inventory against the
reference's counts, freeze regexes, LR schedules, optimizer config."""
import json
import re

import pytest

import oracle as o


def test_attrdict_and_config(tmp_path):
    from retinanet.cfg import AttrDict, Config, default_params
    p = default_params()
    f = tmp_path / "c.json"
    f.write_text(json.dumps(p.to_dict()))
    q = Config(str(f)).params
    assert q.input.input_shape == [640, 640] and q.architecture.head.num_classes == 80
    assert q.training.optimizer.lr_params.schedule_type == "cosine_decay"
    q.inference.mode = "PerClassSoftNMS"
    assert q["inference"]["mode"] == "PerClassSoftNMS"
    assert isinstance(q.anchor_params, AttrDict) and q.anchor_params.scales[0] == 1
    with pytest.raises(AttributeError):
        _ = q.nonexistent


def test_graph_matches_reference_inventory():
    from retinanet.cfg import default_params
    from retinanet.model.builder import ModelBuilder
    from retinanet.model.graph import build_retinanet_graph, init_variables
    g = build_retinanet_graph(default_params())
    v = init_variables(g)
    tr = [k for k in v if g.var_specs[k].get("trainable", True)]
    assert len(tr) == 295 and sum(v[k].numel() for k in tr) == 34389556
    assert len(g.bns) == 102
    rx = ModelBuilder.FREEZE_VARS_REGEX["resnet_initial"]
    fr = [k for k in tr if rx.search(k)]
    assert len(fr) == 33 and sum(v[k].numel() for k in fr) == 225344
    assert "conv2d_10/kernel" in fr and "conv2d_11/kernel" not in fr and not any(k.startswith("fpn") for k in fr)
    head = ModelBuilder.FREEZE_VARS_REGEX["head"]
    assert head.search("box-head/box-head-0-conv2d/kernel") and not head.search("box-head/box-head-prediction-conv2d/kernel")
    # initialisers
    assert float(v["class-head/class-head-prediction-conv2d/bias"][0]) == pytest.approx(-4.59511985, rel=1e-6)
    assert float(v["box-head/box-head-prediction-conv2d/bias"].abs().max()) == 0.0
    zero_gammas = [k for k, b in g.bns.items() if b["gamma_zero"]]
    assert len(zero_gammas) == 16 and all(float(v[k + "/gamma"].abs().max()) == 0.0 for k in zero_gammas)
    assert abs(float(v["box-head/box-head-0-conv2d/kernel"].std()) - 0.01) < 5e-4
    # feature map sizes and the output dict of model/builder.py:94-106
    assert g.tensors["g4b2_out"][:3] == (20, 20, 2048) and g.tensors["fpn_out7"][:3] == (5, 5, 256)
    assert g.tensors[g.outputs["class-predictions"]["3"]] == (80, 80, 720, "f32")
    assert g.tensors[g.outputs["box-predictions"]["7"]] == (5, 5, 36, "f32")


def test_unsupported_configs_fail_loudly():
    from retinanet.cfg import default_params
    from retinanet.model.graph import build_retinanet_graph
    p = default_params()
    p.architecture.backbone.type = "efficientnet-lite3"
    with pytest.raises(NotImplementedError):
        build_retinanet_graph(p)
    p = default_params()
    p.architecture.feature_fusion.type = "bifpn"
    with pytest.raises(ValueError):
        build_retinanet_graph(p)


def test_efficientnet_b3_graph_inventory():
    """B3 features '2'..'5' = 32/48/136/384 channels at strides 4/8/16/32; separable
    FPN/heads at 160 filters; Keras variable names of MBConvBlock._build (efficientnet.py:335-421)."""
    from retinanet.cfg import efficientnet_params
    from retinanet.model.graph import build_retinanet_graph
    from retinanet.model.backbone.efficientnet import block_table, round_filters
    t = block_table("efficientnet-b3")
    assert round_filters(32, 1.2) == 40 and len(t) == 26
    assert [(b["cin"], b["cout"], b["k"], b["stride"]) for b in t[:3]] == [(40, 24, 3, 1), (24, 24, 3, 1), (24, 32, 3, 2)]
    assert (t[-1]["cout"], t[0]["se"], t[2]["se"], t[-1]["se"]) == (384, 10, 6, 96)
    g = build_retinanet_graph(efficientnet_params(input_size=640))
    assert [g.tensors[n][:3] for n in ("b4_out", "b7_out", "b17_out", "b25_out")] == \
        [(160, 160, 32), (80, 80, 48), (40, 40, 136), (20, 20, 384)]
    for name in ("efficientnet-b3/blocks_0/tpu_batch_normalization_1/gamma",      # e1 block: 2 BNs, 1 conv
                 "efficientnet-b3/blocks_0/conv2d/kernel",
                 "efficientnet-b3/blocks_2/conv2d_1/kernel", "efficientnet-b3/blocks_2/tpu_batch_normalization_2/beta",
                 "efficientnet-b3/blocks_2/depthwise_conv2d/depthwise_kernel",
                 "efficientnet-b3/blocks_2/se/conv2d_1/bias", "efficientnet-b3/stem/conv2d/kernel",
                 "fpn/p3-out-conv-3x3/depthwise_kernel", "fpn/p3-out-conv-3x3/pointwise_kernel",
                 "class-head/class-head-prediction-conv2d/pointwise_kernel", "class-head/class-head-prediction-conv2d/bias"):
        assert name in g.var_specs, name
    assert "efficientnet-b3/blocks_0/conv2d_1/kernel" not in g.var_specs
    assert g.var_specs["class-head/class-head-prediction-conv2d/pointwise_kernel"]["shape"] == (1, 1, 160, 720)
    assert g.tensors[g.outputs["class-predictions"]["3"]] == (80, 80, 720, "f32")


def test_lr_schedules_match_oracle():
    from retinanet.cfg import default_params
    from retinanet.optimizers import build_optimizer
    p = default_params()
    opt = build_optimizer(p.training.optimizer, p.training.train_steps, p.floatx.precision)
    for s in (0, 1, 200, 499, 500, 501, 8000, 16374, 16375, 16874):
        assert opt.lr(s) == pytest.approx(o.cosine_decay_with_warmup(s, 0.32, 0.008, 500, 16875, 1e-4), rel=1e-12)
    assert opt.lr(500) == pytest.approx(0.31926448651596, rel=1e-12)   # hand-derived, SURVEY 8(c)
    assert opt.momentum == 0.9 and opt.clipnorm == 10.0 and opt.use_moving_average
    assert opt.ema_decay(0) == pytest.approx(0.1) and opt.ema_decay(10 ** 6) == 0.9998   # min(d, (1+t)/(10+t))
    p.training.optimizer.lr_params = {"schedule_type": "piecewise_constant_decay", "warmup_learning_rate": 0.0067,
                                      "warmup_steps": 500, "boundaries": [1000, 2000], "values": [0.08, 0.008, 0.0008]}
    opt = build_optimizer(p.training.optimizer, 3000, "mixed_bfloat16")
    for s in (0, 250, 499, 500, 998, 999, 1000, 1998, 1999, 2000, 2999):
        assert opt.lr(s) == pytest.approx(o.piecewise_constant_with_warmup(s, 0.0067, 500, [1000, 2000], [0.08, 0.008, 0.0008]))
    assert opt.lr(999) == 0.08 and opt.lr(1000) == 0.008     # boundaries shifted by -1 (piecewise...py:8-9)


def test_pixel_pair_kernel_is_the_same_convolution():
    """engine.pixel_pair_kernel: a 3x3 / stride 1 / pad 1 convolution on [N, H, W, C] equals the convolution with the paired
    kernel on the same bytes seen as [N, H, W/2, 2C] (two adjacent pixels = one pixel of 2C channels) — checked against
    torch's CPU conv in float64, including the left / right image borders (a pair column that is half padding)."""
    import torch
    import torch.nn.functional as F
    from retinanet.model.engine import pixel_pair_kernel
    g = torch.Generator().manual_seed(4)
    for (N, H, W, C) in ((2, 5, 8, 3), (1, 4, 6, 8), (1, 1, 2, 4)):
        x = torch.randn((N, H, W, C), generator=g, dtype=torch.float64)
        w = torch.randn((3, 3, C, C), generator=g, dtype=torch.float64)        # HWIO
        want = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
        w2 = pixel_pair_kernel(w)
        assert w2.shape == (3, 3, 2 * C, 2 * C)
        assert int((w2 != 0).sum()) == int((w != 0).sum()) * 2                 # every tap once per pixel of the pair
        x2 = x.reshape(N, H, W // 2, 2 * C)
        got = F.conv2d(x2.permute(0, 3, 1, 2), w2.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1).reshape(N, H, W, C)
        torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-12)


def test_bottleneck_blocks_are_found_structurally(monkeypatch):
    """retinanet/model/bottleneck.py::find_blocks: the three 64-wide stride-1 bottleneck blocks of ResNet-50's first group
    (resnet.py:194-248, :324-331) — the projection block with Cx = 64, two identity blocks with Cx = 256 — and nothing
    else: not the 128-wide stages, no block of EfficientNet; RNET_FUSE_BOTTLENECK=0 switches the recognition off"""
    from retinanet.cfg import default_params, efficientnet_params
    from retinanet.model.bottleneck import find_blocks
    from retinanet.model.graph import build_retinanet_graph
    g = build_retinanet_graph(default_params(input_size=640))
    blocks = find_blocks(g)
    assert [b["name"] for b in blocks] == ["g1b0_out", "g1b1_out", "g1b2_out"]
    assert [b["Cx"] for b in blocks] == [64, 256, 256]
    assert [o["out"] for o in blocks[0]["ops"]] == ["g1b0_sc", "g1b0_a", "g1b0_b", "g1b0_out"]
    assert blocks[0]["sc"] is not None and blocks[1]["sc"] is None and blocks[1]["x"] == "g1b0_out"
    assert find_blocks(build_retinanet_graph(efficientnet_params("efficientnet-b3", input_size=640))) == []
    monkeypatch.setenv("RNET_FUSE_BOTTLENECK", "0")
    assert find_blocks(g) == []
