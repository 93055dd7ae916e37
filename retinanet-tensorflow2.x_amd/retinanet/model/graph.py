"""Static layer graph of the RetinaNet (ResNet + FPN + shared heads) and its variables.

This is the host-side description the executors in `retinanet.model.engine` turn into HIP
launches; it plays the role of the Keras functional graph the reference builds in
retinanet/model/builder.py:36-106 (backbone resnet.py:289-341, neck fpn_base.py:54-71 +
fpn.py:81-107, BalanceFeatures balance_features.py:19-60, heads detection_head.py:90-104).
Variable names follow the Keras names so FREEZE_VARS_REGEX and weight
files keyed by name keep working; conv kernels are HWIO float32 like the reference's.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3], 200: [3, 24, 36, 3],
                  26: [2, 2, 2, 2], 14: [1, 1, 1, 1]}


class Graph:
    """ops: list of dicts; tensors: name -> (H, W, C, dtype) with an implicit batch dim."""

    def __init__(self):
        self.tensors = OrderedDict()
        self.ops = []
        self.var_specs = OrderedDict()  # name -> dict(shape, init, **kw)
        self.convs = OrderedDict()      # conv layer name -> dict(k, cin, cout, stride, bias)
        self.bns = OrderedDict()        # bn layer name -> dict(C, gamma_zero)
        self.outputs = {}

    def tensor(self, name, H, W, C, dtype="bf16"):
        self.tensors[name] = (H, W, C, dtype)
        return name

    def add_conv_layer(self, name, k, cin, cout, stride, bias, init, bias_init=0.0):
        self.convs[name] = dict(k=k, cin=cin, cout=cout, stride=stride, bias=bias)
        self.var_specs[name + "/kernel"] = dict(shape=(k, k, cin, cout), init=init)
        if bias:
            self.var_specs[name + "/bias"] = dict(shape=(cout,), init="const", value=bias_init)

    def add_bn_layer(self, name, C, gamma_zero=False):
        self.bns[name] = dict(C=C, gamma_zero=gamma_zero)
        self.var_specs[name + "/gamma"] = dict(shape=(C,), init="const", value=0.0 if gamma_zero else 1.0)
        self.var_specs[name + "/beta"] = dict(shape=(C,), init="const", value=0.0)
        self.var_specs[name + "/moving_mean"] = dict(shape=(C,), init="const", value=0.0, trainable=False)
        self.var_specs[name + "/moving_variance"] = dict(shape=(C,), init="const", value=1.0, trainable=False)

    def conv(self, out, inp, conv, bn=None, act=None, residual=None, group=None, out_dtype="bf16", pad=None):
        c = self.convs[conv]
        H, W, C, _ = self.tensors[inp]
        assert C == c["cin"], (conv, C, c["cin"])
        k, s = c["k"], c["stride"]
        if pad is None:
            pad = (k - 1) // 2  # stride 1 SAME, or fixed_padding + VALID (resnet.py:103-141)
        Ho = (H + 2 * pad - k) // s + 1
        Wo = (W + 2 * pad - k) // s + 1
        self.tensor(out, Ho, Wo, c["cout"], out_dtype)
        self.ops.append(dict(op="conv", out=out, inp=inp, conv=conv, bn=bn, act=act, residual=residual,
                             group=group, out_dtype=out_dtype, pad=pad))
        return out


def _bn_name(i, sync):
    base = "sync_batch_normalization" if sync else "batch_normalization"
    return base if i == 0 else f"{base}_{i}"


def _conv_name(i):
    return "conv2d" if i == 0 else f"conv2d_{i}"


def build_retinanet_graph(params, sync_bn_names=False):
    arch = params.architecture
    if "resnet" not in arch.backbone.type.lower():
        raise NotImplementedError(f"backbone {arch.backbone.type}: only the ResNet family is built so far "
                                  "(EfficientNet-B3 is SURVEY §8 row a18, next)")
    depth = int(arch.backbone.depth)
    if depth not in _RESNET_LAYERS:
        raise ValueError(f"unsupported bottleneck ResNet depth {depth}")
    if arch.conv_2d.use_seperable_conv:
        raise NotImplementedError("use_seperable_conv is only used by the EfficientNet/MobileDet configs")
    if arch.feature_fusion.type != "fpn":
        raise ValueError("{} FPN not implemented".format(arch.feature_fusion.type))
    if arch.feature_fusion.fusion_mode != "sum":
        raise NotImplementedError("fusion_mode other than 'sum' is unused by every shipped config")
    if arch.auxillary_head.use_auxillary_head:
        raise NotImplementedError("auxillary head is disabled in every shipped config")
    H, W = params.input.input_shape
    g = Graph()
    act = arch.activation.type
    cidx = [0]

    def rconv(k, cin, cout, stride):
        name = _conv_name(cidx[0])
        g.add_conv_layer(name, k, cin, cout, stride, bias=False, init="variance_scaling")
        return name

    def rbn(C, zero=False):
        name = _bn_name(cidx[0], sync_bn_names)
        g.add_bn_layer(name, C, gamma_zero=zero)
        cidx[0] += 1
        return name

    # ---- ResNet (resnet.py:289-341); ResNet blocks always use ReLU (resnet.py:68-69) ---------
    g.tensor("images", H, W, 3, "f32")
    c = rconv(7, 3, 64, 2)
    b = rbn(64)
    Hs, Ws = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    g.tensor("stem", Hs, Ws, 64)
    g.ops.append(dict(op="stem", out="stem", inp="images", conv=c, bn=b, act="relu"))
    Hp, Wp = math.ceil(Hs / 2), math.ceil(Ws / 2)
    # MaxPool 3x3 s2 SAME: total pad = max((Ho-1)*2+3-H, 0), before = total//2 (TF rule)
    pt = max((Hp - 1) * 2 + 3 - Hs, 0) // 2
    pl = max((Wp - 1) * 2 + 3 - Ws, 0) // 2
    g.tensor("pool", Hp, Wp, 64)
    g.ops.append(dict(op="maxpool", out="pool", inp="stem", k=3, stride=2, pad_top=pt, pad_left=pl))
    x, cin = "pool", 64
    feats = {}
    for gi, (filters, blocks, stride) in enumerate(zip([64, 128, 256, 512], _RESNET_LAYERS[depth], [1, 2, 2, 2])):
        for bi in range(blocks):
            s = stride if bi == 0 else 1
            pre = f"g{gi + 1}b{bi}"
            shortcut = x
            if bi == 0:
                pc = rconv(1, cin, 4 * filters, s)
                pb = rbn(4 * filters)
                shortcut = g.conv(pre + "_sc", x, pc, pb, act=None, pad=0)
            c1 = rconv(1, cin, filters, 1)
            b1 = rbn(filters)
            t = g.conv(pre + "_a", x, c1, b1, act="relu")
            c2 = rconv(3, filters, filters, s)
            b2 = rbn(filters)
            t = g.conv(pre + "_b", t, c2, b2, act="relu")
            c3 = rconv(1, filters, 4 * filters, 1)
            b3 = rbn(4 * filters, zero=True)
            x = g.conv(pre + "_out", t, c3, b3, act="relu", residual=shortcut)
            cin = 4 * filters
        feats[str(gi + 2)] = x

    # ---- FPN (fpn_base.py:54-71, fpn.py:81-107) ------------------------------------------------
    ff = arch.feature_fusion
    F = int(ff.filters)
    lo, hi, bmax = int(ff.min_level), int(ff.max_level), int(ff.backbone_max_level)
    bn_tag = "sync_batch_normalization" if sync_bn_names else "batch_normalization"
    top = feats[str(bmax)]
    ctop = g.tensors[top][2]
    g.add_conv_layer("fpn/backbone_max_level_conv_1x1", 1, ctop, F, 1, bias=True, init="variance_scaling")
    g.add_bn_layer(f"fpn/backbone_max_level_{bn_tag}", F)
    g.conv("fpn_c6pre", top, "fpn/backbone_max_level_conv_1x1", f"fpn/backbone_max_level_{bn_tag}", act=None,
           group="fpn_1x1")
    prev = "fpn_c6pre"
    for level in range(bmax + 1, hi + 1):
        Hl, Wl = g.tensors[prev][0] // 2, g.tensors[prev][1] // 2
        g.tensor(f"fpn_in{level}", Hl, Wl, F)
        g.ops.append(dict(op="maxpool", out=f"fpn_in{level}", inp=prev, k=2, stride=2, pad_top=0, pad_left=0))
        prev = f"fpn_in{level}"
    for level in range(lo, bmax + 1):
        src = feats[str(level)]
        name = f"fpn/p{level}-in-channel-normalize-conv-1x1"
        bn = f"fpn/p{level}-in-channel-normalize-{bn_tag}"
        g.add_conv_layer(name, 1, g.tensors[src][2], F, 1, bias=True, init="variance_scaling")
        g.add_bn_layer(bn, F)
        g.conv(f"fpn_in{level}", src, name, bn, act=None, group="fpn_1x1")
    levels = list(range(lo, hi + 1))
    for level in levels[:-1]:
        Hl, Wl, _, _ = g.tensors[f"fpn_in{level}"]
        g.tensor(f"fpn_td{level}", Hl, Wl, F)
    g.ops.append(dict(op="topdown", ins=[f"fpn_in{l}" for l in levels],
                      outs=[f"fpn_td{l}" for l in levels[:-1]] + [f"fpn_in{hi}"], act=act))
    for level in levels:
        name = f"fpn/p{level}-out-conv-3x3"
        bn = f"fpn/p{level}-out-{bn_tag}"
        g.add_conv_layer(name, 3, F, F, 1, bias=True, init="variance_scaling")
        g.add_bn_layer(bn, F)
        src = f"fpn_td{level}" if level != hi else f"fpn_in{hi}"
        g.conv(f"fpn_out{level}", src, name, bn, act=None, group="fpn_out")
    feat = {l: f"fpn_out{l}" for l in levels}
    if ff.use_balanced_features:
        g.ops.append(dict(op="balance", tensors=[feat[l] for l in levels], mid=1))  # min_level + 1

    # ---- heads (detection_head.py:8-104, head/builder.py:7-43) ------------------------------
    hd = arch.head
    nconv, HF = int(hd.num_convs), int(hd.filters)
    A, K = int(hd.num_anchors), int(hd.num_classes)
    outs = {"box": {}, "class": {}}
    for head, ofilt, bias_init in (("box-head", A * 4, 0.0),
                                   ("class-head", A * K, -float(np.log((1 - 0.01) / 0.01)))):
        for i in range(nconv):
            g.add_conv_layer(f"{head}/{head}-{i}-conv2d", 3, F if i == 0 else HF, HF, 1, bias=True,
                             init="normal_0.01")
            for level in levels:
                g.add_bn_layer(f"{head}/{head}-{i}-p{level}-{bn_tag}", HF)
        g.add_conv_layer(f"{head}/{head}-prediction-conv2d", 3, HF, ofilt, 1, bias=True, init="normal_0.01",
                         bias_init=bias_init)
    for i in range(nconv):
        for head in ("box-head", "class-head"):
            for level in levels:
                src = feat[level] if i == 0 else f"{head}_t{i - 1}_p{level}"
                g.conv(f"{head}_t{i}_p{level}", src, f"{head}/{head}-{i}-conv2d",
                       f"{head}/{head}-{i}-p{level}-{bn_tag}", act=act, group=f"tower{i}")
    for head, key in (("box-head", "box"), ("class-head", "class")):
        for level in levels:
            src = f"{head}_t{nconv - 1}_p{level}" if nconv else feat[level]
            g.conv(f"{head}_pred_p{level}", src, f"{head}/{head}-prediction-conv2d", None, act=None,
                   group=f"pred_{key}", out_dtype="f32")
            outs[key][str(level)] = f"{head}_pred_p{level}"
    g.outputs = {"class-predictions": outs["class"], "box-predictions": outs["box"]}
    g.levels = levels
    g.meta = dict(num_anchors=A, num_classes=K, filters=F)
    return g


def init_variables(graph, seed=1337, device="cpu"):
    """Reference initialisers: VarianceScaling() for ResNet/FPN kernels
    (resnet.py:143, fpn_base.py:30-33), N(0, 0.01) for head kernels (detection_head.py:40-43),
    class prediction bias -log(99) (head/builder.py:30), last block BN gamma 0 (resnet.py:246)."""
    import torch
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    out = OrderedDict()
    for name, spec in graph.var_specs.items():
        shape = spec["shape"]
        if spec["init"] == "const":
            t = torch.full(shape, float(spec["value"]), dtype=torch.float32)
        elif spec["init"] == "variance_scaling":
            fan_in = shape[0] * shape[1] * shape[2]
            std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
            t = torch.empty(shape, dtype=torch.float32)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
        elif spec["init"] == "normal_0.01":
            t = torch.empty(shape, dtype=torch.float32).normal_(0.0, 0.01, generator=gen)
        else:
            raise ValueError(spec["init"])
        out[name] = t.to(device)
    return out
