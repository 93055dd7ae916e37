"""EfficientNet-B0..B8 feature extractor as static graph ops.

Mirrors what retinanet/model/backbone/efficientnet.py builds for `features_only=True`
(`Model._build` :699-781, `Model.call` :783-855, `MBConvBlock` :291-482, `SE` :222-265,
`Stem` :566-586, block strings :82-90, `round_filters`/`round_repeats` :196-219) and returns the
'2'..'5' feature map names (`EfficientNet.__init__` :1033-1036: reduction_2..reduction_5).

drop_connect (:97-113) is the identity when training=False; in training the project conv of every
skip block carries its `survival` probability and the TrainEngine applies a per-image 0 / (1/p) factor to
the BatchNorm output before the residual add.  The classification Head (:589-673) is never built into
the detector.  Variable names follow the
Keras scopes `<model>/stem/...`, `<model>/blocks_<i>/...` with the conv2d / tpu_batch_normalization
counters of `MBConvBlock._build` (:335-421).
"""
from __future__ import annotations

import math

_PARAMS = {  # efficientnet.py:48-62: (width_coefficient, depth_coefficient)
    "efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1), "efficientnet-b2": (1.1, 1.2),
    "efficientnet-b3": (1.2, 1.4), "efficientnet-b4": (1.4, 1.8), "efficientnet-b5": (1.6, 2.2),
    "efficientnet-b6": (1.8, 2.6), "efficientnet-b7": (2.0, 3.1), "efficientnet-b8": (2.2, 3.6),
}

# efficientnet.py:82-90 decoded: (repeats, kernel, stride, expand, in, out, se_ratio)
_BLOCKS = [(1, 3, 1, 1, 32, 16, 0.25), (2, 3, 2, 6, 16, 24, 0.25), (2, 5, 2, 6, 24, 40, 0.25),
           (3, 3, 2, 6, 40, 80, 0.25), (3, 5, 1, 6, 80, 112, 0.25), (4, 5, 2, 6, 112, 192, 0.25),
           (1, 3, 1, 6, 192, 320, 0.25)]


SURVIVAL_PROB = 0.8   # efficientnet.py:940 (`survival_prob=0.8` for every efficientnet-b* / lite model)


def round_filters(filters, width, divisor=8):
    """efficientnet.py:196-211."""
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def round_repeats(repeats, depth):
    """efficientnet.py:214-219."""
    return int(math.ceil(depth * repeats))


def block_table(model_name):
    """Per-block (kernel, stride, expand, cin, cout, se_filters) list after scaling (:715-779)."""
    width, depth = _PARAMS[model_name]
    out = []
    for (rep, k, s, e, cin, cout, se) in _BLOCKS:
        cin, cout = round_filters(cin, width), round_filters(cout, width)
        for r in range(round_repeats(rep, depth)):
            bi, bs = (cin, s) if r == 0 else (cout, 1)
            out.append(dict(k=k, stride=bs, expand=e, cin=bi, cout=cout, se=max(1, int(bi * se))))
    return out


def build_efficientnet_backbone(g, model_name, H, W, sync_bn_names=False):
    if model_name not in _PARAMS:
        raise NotImplementedError(f"model name is not pre-defined: {model_name}")
    width, _ = _PARAMS[model_name]
    if H % 2 or W % 2:
        raise ValueError("EfficientNet stem packing expects even input sizes")
    pre = model_name + "/"
    bn_base = "tpu_batch_normalization"
    # the Stem's conv and BN are unnamed (:571-581), so Keras auto-names them
    stem_bn = pre + "stem/" + ("sync_batch_normalization" if sync_bn_names else "batch_normalization")

    stem_c = round_filters(32, width)
    g.add_conv_layer(pre + "stem/conv2d", 3, 3, stem_c, 2, bias=False, init="effnet_conv")
    g.add_bn_layer(stem_bn, stem_c)
    g.tensor("stem", H // 2, W // 2, stem_c)
    g.ops.append(dict(op="stem", out="stem", inp="images", conv=pre + "stem/conv2d", bn=stem_bn,
                      act="swish", k=3, pad_top=0, pad_left=0))
    x = "stem"
    blocks = block_table(model_name)
    reductions = {}
    ridx = 0
    for i, b in enumerate(blocks):
        scope = f"{pre}blocks_{i}/"
        nconv, nbn = [0], [0]

        def conv_name():
            n = "conv2d" if nconv[0] == 0 else f"conv2d_{nconv[0]}"
            nconv[0] += 1
            return scope + n

        def bn_name():
            n = bn_base if nbn[0] == 0 else f"{bn_base}_{nbn[0]}"
            nbn[0] += 1
            return scope + n

        inp = x
        cexp = b["cin"] * b["expand"]
        if b["expand"] != 1:
            cn, bn = conv_name(), bn_name()
            g.add_conv_layer(cn, 1, b["cin"], cexp, 1, bias=False, init="effnet_conv")
            g.add_bn_layer(bn, cexp)
            x = g.conv(f"b{i}_expand", x, cn, bn, act="swish")
        dn, bn = scope + "depthwise_conv2d", bn_name()
        g.add_dw_layer(dn, b["k"], cexp, b["stride"], "effnet_conv")
        g.add_bn_layer(bn, cexp)
        x = g.dwconv(f"b{i}_dw", x, dn, bn=bn, act="swish")
        g.add_se_layer(scope + "se", cexp, b["se"])
        g.ops.append(dict(op="se", tensor=x, se=scope + "se"))
        cn, bn = conv_name(), bn_name()
        g.add_conv_layer(cn, 1, cexp, b["cout"], 1, bias=False, init="effnet_conv")
        g.add_bn_layer(bn, b["cout"])
        skip = inp if (b["stride"] == 1 and b["cin"] == b["cout"]) else None
        x = g.conv(f"b{i}_out", x, cn, bn, act=None, residual=skip)
        if skip is not None:
            # drop_connect in training (:97-113, :824-827): survival_prob 0.8 scaled linearly with the block index
            g.ops[-1]["survival"] = 1.0 - (1.0 - SURVIVAL_PROB) * float(i) / len(blocks)
        # efficientnet.py:814-817: a block is a reduction point when it is the last one or the next
        # block strides
        if i == len(blocks) - 1 or blocks[i + 1]["stride"] > 1:
            ridx += 1
            reductions[ridx] = x
    return {str(l): reductions[l] for l in range(2, 6)}


class EfficientNet:
    """`EfficientNet(input_shape, model_name, checkpoint, normalization_op_params, override_params)`
    (efficientnet.py:1003-1040): features '2'..'5' = reduction_2..reduction_5."""

    def __init__(self, input_shape, model_name, checkpoint="", normalization_op_params=None, override_params=None):
        if not model_name.startswith("efficientnet-b"):
            raise NotImplementedError(f"backbone {model_name}: the EfficientNet-B family is built; EfficientNet-lite "
                                      "is out of scope (SURVEY §2.1)")
        if override_params:
            raise NotImplementedError("efficientnet override_params are unused by every shipped config")
        self.input_shape, self.model_name, self.checkpoint = list(input_shape), model_name, checkpoint
        self._sync_names = bool((normalization_op_params or {}).get("sync_names", False))
        self.name = model_name

    def __call__(self, images):
        from retinanet.model.graph import Sym
        g = images.graph
        H, W, _, _ = g.tensors[images.name]
        feats = build_efficientnet_backbone(g, self.model_name, H, W, self._sync_names)
        return {lv: Sym(g, n) for lv, n in feats.items()}
