"""Static layer graph of the RetinaNet (ResNet + FPN + shared heads) and its variables.

This is the host-side description the executors in `retinanet.model.engine` turn into HIP
launches; it plays the role of the Keras functional graph the reference builds in
retinanet/model/builder.py:36-106.  The pieces are added by the sub-builders, named like the reference's:
`build_backbone` (model/backbone/), `build_neck` (model/neck/), `build_detection_heads` (model/head/),
`BalanceFeatures` (model/layers/) — each returns a layer object that is CALLED on symbolic tensors (`Sym`).
Variable names follow the Keras names so FREEZE_VARS_REGEX and weight
files keyed by name keep working; conv kernels are HWIO float32 like the reference's.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

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


class Sym:
    """A symbolic tensor: a name in a `Graph` — what calling a sub-builder's layer on an input returns, the way
    calling a Keras layer on a `tf.keras.Input` returns a KerasTensor (model/builder.py:47-93)."""

    def __init__(self, graph, name):
        self.graph, self.name = graph, name

    @property
    def shape(self):
        H, W, C, _ = self.graph.tensors[self.name]
        return (None, H, W, C)

    def __repr__(self):
        return f"Sym({self.name}, {self.shape})"


def graph_input(input_shape, name="images"):
    """tf.keras.Input(shape=[H, W, channels], name='images') (model/builder.py:46-50): a fresh graph and its input"""
    H, W, C = input_shape
    g = Graph()
    g.tensor(name, H, W, C, "f32")
    return Sym(g, name)


def _bn_name(i, sync):
    base = "sync_batch_normalization" if sync else "batch_normalization"
    return base if i == 0 else f"{base}_{i}"


def _conv_name(i):
    return "conv2d" if i == 0 else f"conv2d_{i}"


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


def build_retinanet_graph(params, sync_bn_names=False):
    """The whole detector as one static graph, composed from the sub-builders exactly as the reference's
    ModelBuilder.__call__ composes its Keras layers (model/builder.py:36-106)."""
    from retinanet.model.backbone import build_backbone
    from retinanet.model.head import build_detection_heads
    from retinanet.model.layers.balance_features import BalanceFeatures
    from retinanet.model.neck import build_neck
    from retinanet.model.utils import get_activation_op
    arch = params.architecture
    if arch.auxillary_head.use_auxillary_head:
        raise NotImplementedError("auxillary head is disabled in every shipped config")
    norm = dict(arch.batch_norm)
    norm["sync_names"] = bool(sync_bn_names)
    input_shape = list(params.input.input_shape) + [int(params.input.get("channels", 3))]
    images = graph_input(input_shape)
    activation_fn = get_activation_op(arch.activation.type)
    backbone = build_backbone(input_shape=input_shape, params=arch.backbone, normalization_op_params=norm)
    neck = build_neck(params=arch.feature_fusion, conv_2d_op_params=arch.conv_2d, normalization_op_params=norm,
                      activation_fn=activation_fn)
    box_head, class_head = build_detection_heads(
        params=arch.head, min_level=arch.feature_fusion.min_level, max_level=arch.feature_fusion.max_level,
        conv_2d_op_params=arch.conv_2d, normalization_op_params=norm, activation_fn=activation_fn)
    features = neck(backbone(images))
    if arch.feature_fusion.use_balanced_features:
        features = BalanceFeatures(min_level=arch.feature_fusion.min_level, max_level=arch.feature_fusion.max_level,
                                   intermediate_level=arch.feature_fusion.min_level + 1)(features)
    box_outputs = box_head(features)
    class_outputs = class_head(features)
    g = images.graph
    g.outputs = {"class-predictions": {lv: t.name for lv, t in class_outputs.items()},
                 "box-predictions": {lv: t.name for lv, t in box_outputs.items()}}
    g.levels = list(range(int(arch.feature_fusion.min_level), int(arch.feature_fusion.max_level) + 1))
    g.meta = dict(num_anchors=int(arch.head.num_anchors), num_classes=int(arch.head.num_classes),
                  filters=int(arch.feature_fusion.filters))
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
