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

    def add_conv_layer(self, name, k, cin, cout, stride, bias, init, bias_init=0.0, kernel_var=None):
        kv = kernel_var or (name + "/kernel")
        self.convs[name] = dict(k=k, cin=cin, cout=cout, stride=stride, bias=bias, kvar=kv, bvar=name + "/bias")
        self.var_specs[kv] = dict(shape=(k, k, cin, cout), init=init)
        if bias:
            self.var_specs[name + "/bias"] = dict(shape=(cout,), init="const", value=bias_init)

    def add_dw_layer(self, name, k, C, stride, init, kernel_var=None):
        """DepthwiseConv2D / the depthwise half of SeparableConv2D: kernel [k,k,C,1]."""
        kv = kernel_var or (name + "/depthwise_kernel")
        self.dws = getattr(self, "dws", OrderedDict())
        self.dws[name] = dict(k=k, C=C, stride=stride, kvar=kv)
        self.var_specs[kv] = dict(shape=(k, k, C, 1), init=init)

    def add_se_layer(self, name, C, se):
        self.ses = getattr(self, "ses", OrderedDict())
        self.ses[name] = dict(C=C, se=se)
        self.var_specs[name + "/conv2d/kernel"] = dict(shape=(1, 1, C, se), init="effnet_conv")
        self.var_specs[name + "/conv2d/bias"] = dict(shape=(se,), init="const", value=0.0)
        self.var_specs[name + "/conv2d_1/kernel"] = dict(shape=(1, 1, se, C), init="effnet_conv")
        self.var_specs[name + "/conv2d_1/bias"] = dict(shape=(C,), init="const", value=0.0)

    def dwconv(self, out, inp, dw, bn=None, act=None, group=None):
        d = self.dws[dw]
        H, W, C, _ = self.tensors[inp]
        assert C == d["C"], (dw, C, d["C"])
        k, s = d["k"], d["stride"]
        Ho, Wo = -(-H // s), -(-W // s)
        # TF SAME: total = max((Ho-1)*s + k - H, 0), before = total // 2
        pt = max((Ho - 1) * s + k - H, 0) // 2
        pl = max((Wo - 1) * s + k - W, 0) // 2
        self.tensor(out, Ho, Wo, C)
        self.ops.append(dict(op="dwconv", out=out, inp=inp, dw=dw, bn=bn, act=act, group=group, pad_top=pt,
                             pad_left=pl))
        return out

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
    btype = arch.backbone.type.lower()
    if "resnet" not in btype and not btype.startswith("efficientnet-b"):
        raise NotImplementedError(f"backbone {arch.backbone.type}: the ResNet and EfficientNet-B families are built; "
                                  "EfficientNet-lite / MobileDet are out of scope (SURVEY §2.1)")
    if "resnet" in btype:
        depth = int(arch.backbone.depth)
        if depth not in _RESNET_LAYERS:
            raise ValueError(f"unsupported bottleneck ResNet depth {depth}")
    separable = bool(arch.conv_2d.use_seperable_conv)
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

    g.tensor("images", H, W, 3, "f32")
    if btype.startswith("efficientnet-b"):
        from retinanet.model.graph_efficientnet import build_efficientnet_backbone
        feats = build_efficientnet_backbone(g, btype, H, W, sync_bn_names)
        return _build_fpn_and_heads(g, params, feats, act, sync_bn_names, separable)
    # ---- ResNet (resnet.py:289-341); ResNet blocks always use ReLU (resnet.py:68-69) ---------
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
    return _build_fpn_and_heads(g, params, feats, act, sync_bn_names, separable)


def _conv_or_sep(g, separable, out, inp, name, k, cout, bias_init, init, bn, act, group, out_dtype="bf16",
                 define=True):
    """tf.keras.layers.Conv2D or SeparableConv2D (fpn_base.py:28-39, detection_head.py:37-50): the
    separable form is a depthwise k x k (no bias, no activation) followed by a pointwise 1x1 + bias."""
    cin = g.tensors[inp][2]
    if not separable:
        if define:
            g.add_conv_layer(name, k, cin, cout, 1, bias=True, init=init, bias_init=bias_init)
        return g.conv(out, inp, name, bn, act=act, group=group, out_dtype=out_dtype)
    if define:
        g.add_dw_layer(name + ":dw", k, cin, 1, "variance_scaling", kernel_var=name + "/depthwise_kernel")
        g.add_conv_layer(name, 1, cin, cout, 1, bias=True, init="variance_scaling", bias_init=bias_init,
                         kernel_var=name + "/pointwise_kernel")
    g.dwconv(out + ":dw", inp, name + ":dw", bn=None, act=None, group=(group + ":dw") if group else None)
    return g.conv(out, out + ":dw", name, bn, act=act, group=group, out_dtype=out_dtype, pad=0)


def _build_fpn_and_heads(g, params, feats, act, sync_bn_names, separable):
    arch = params.architecture
    # ---- FPN (fpn_base.py:54-71, fpn.py:81-107) ------------------------------------------------
    ff = arch.feature_fusion
    F = int(ff.filters)
    lo, hi, bmax = int(ff.min_level), int(ff.max_level), int(ff.backbone_max_level)
    bn_tag = "sync_batch_normalization" if sync_bn_names else "batch_normalization"
    top = feats[str(bmax)]
    ctop = g.tensors[top][2]
    g.add_bn_layer(f"fpn/backbone_max_level_{bn_tag}", F)
    _conv_or_sep(g, separable, "fpn_c6pre", top, "fpn/backbone_max_level_conv_1x1", 1, F, 0.0, "variance_scaling",
                 f"fpn/backbone_max_level_{bn_tag}", None, None if separable else "fpn_1x1")
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
        g.add_bn_layer(bn, F)
        _conv_or_sep(g, separable, f"fpn_in{level}", src, name, 1, F, 0.0, "variance_scaling", bn, None,
                     None if separable else "fpn_1x1")
    levels = list(range(lo, hi + 1))
    for level in levels[:-1]:
        Hl, Wl, _, _ = g.tensors[f"fpn_in{level}"]
        g.tensor(f"fpn_td{level}", Hl, Wl, F)
    g.ops.append(dict(op="topdown", ins=[f"fpn_in{l}" for l in levels],
                      outs=[f"fpn_td{l}" for l in levels[:-1]] + [f"fpn_in{hi}"], act=act))
    for level in levels:
        name = f"fpn/p{level}-out-conv-3x3"
        bn = f"fpn/p{level}-out-{bn_tag}"
        g.add_bn_layer(bn, F)
        src = f"fpn_td{level}" if level != hi else f"fpn_in{hi}"
        _conv_or_sep(g, separable, f"fpn_out{level}", src, name, 3, F, 0.0, "variance_scaling", bn, None,
                     "fpn_out")
    feat = {l: f"fpn_out{l}" for l in levels}
    if ff.use_balanced_features:
        g.ops.append(dict(op="balance", tensors=[feat[l] for l in levels], mid=1))  # min_level + 1

    # ---- heads (detection_head.py:8-104, head/builder.py:7-43) ------------------------------
    hd = arch.head
    nconv, HF = int(hd.num_convs), int(hd.filters)
    A, K = int(hd.num_anchors), int(hd.num_classes)
    outs = {"box": {}, "class": {}}
    head_init = "variance_scaling" if separable else "normal_0.01"
    for head in ("box-head", "class-head"):
        for i in range(nconv):
            for level in levels:
                g.add_bn_layer(f"{head}/{head}-{i}-p{level}-{bn_tag}", HF)
    defined = set()
    for i in range(nconv):
        for head in ("box-head", "class-head"):
            for level in levels:
                src = feat[level] if i == 0 else f"{head}_t{i - 1}_p{level}"
                name = f"{head}/{head}-{i}-conv2d"
                _conv_or_sep(g, separable, f"{head}_t{i}_p{level}", src, name, 3, HF, 0.0, head_init,
                             f"{head}/{head}-{i}-p{level}-{bn_tag}", act, f"tower{i}", define=name not in defined)
                defined.add(name)
    for head, key, ofilt, bias_init in (("box-head", "box", A * 4, 0.0),
                                       ("class-head", "class", A * K, -float(np.log((1 - 0.01) / 0.01)))):
        for level in levels:
            src = f"{head}_t{nconv - 1}_p{level}" if nconv else feat[level]
            name = f"{head}/{head}-prediction-conv2d"
            _conv_or_sep(g, separable, f"{head}_pred_p{level}", src, name, 3, ofilt, bias_init, head_init, None, None,
                         f"pred_{key}", out_dtype="f32", define=name not in defined)
            defined.add(name)
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
        elif spec["init"] == "effnet_conv":   # efficientnet.py:116-140: N(0, sqrt(2 / fan_out))
            fan_out = shape[0] * shape[1] * shape[3]
            t = torch.empty(shape, dtype=torch.float32).normal_(0.0, math.sqrt(2.0 / fan_out), generator=gen)
        elif spec["init"] == "normal_0.01":
            t = torch.empty(shape, dtype=torch.float32).normal_(0.0, 0.01, generator=gen)
        else:
            raise ValueError(spec["init"])
        out[name] = t.to(device)
    return out
