"""Bottleneck ResNet as static graph ops — what retinanet/model/backbone/resnet.py builds: `conv2d_fixed_padding`
(:118-144: explicit (k-1)//2 padding + VALID for strided convs), `NormActivation` (:28-79), `bottleneck_block`
(:194-248: stride in the 3x3, projection shortcut 1x1 + BN without activation, last BN gamma zero-initialised),
`block_group` (:251-286), `resnet_fn` (:289-341: 7x7/2 stem, 3x3/2 SAME max-pool, four groups), depths from
`ResNet._MODEL_CONFIG` (:366-369).  Layers are auto-named like Keras names them (`conv2d`, `conv2d_1`, ...,
`batch_normalization[_N]`) so FREEZE_VARS_REGEX and weight files keyed by name keep working; ResNet blocks always
use ReLU (:68-69)."""
from __future__ import annotations

import math

from retinanet.model.graph import Sym, _bn_name, _conv_name

_MODEL_CONFIG = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3], 200: [3, 24, 36, 3],
                 # shallow bottleneck variants of the same builder: parity tests on a well-conditioned net
                 26: [2, 2, 2, 2], 14: [1, 1, 1, 1]}


class ResNet:
    def __init__(self, input_shape, depth=50, checkpoint="", normalization_op_params=None, **_):
        depth = int(depth)
        if depth not in _MODEL_CONFIG:
            raise ValueError(f"unsupported bottleneck ResNet depth {depth}")
        self.input_shape, self.depth, self.checkpoint = list(input_shape), depth, checkpoint
        self._sync_names = bool((normalization_op_params or {}).get("sync_names", False))
        self.name = f"resnet_{depth}"

    def __call__(self, images):
        g, sync = images.graph, self._sync_names
        H, W, _, _ = g.tensors[images.name]
        cidx = [0]

        def rconv(k, cin, cout, stride):
            name = _conv_name(cidx[0])
            g.add_conv_layer(name, k, cin, cout, stride, bias=False, init="variance_scaling")
            return name

        def rbn(C, zero=False):
            name = _bn_name(cidx[0], sync)
            g.add_bn_layer(name, C, gamma_zero=zero)
            cidx[0] += 1
            return name

        c = rconv(7, 3, 64, 2)
        b = rbn(64)
        Hs, Ws = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        g.tensor("stem", Hs, Ws, 64)
        g.ops.append(dict(op="stem", out="stem", inp=images.name, conv=c, bn=b, act="relu"))
        Hp, Wp = math.ceil(Hs / 2), math.ceil(Ws / 2)
        # MaxPool 3x3 s2 SAME: total pad = max((Ho-1)*2+3-H, 0), before = total//2 (TF rule)
        pt = max((Hp - 1) * 2 + 3 - Hs, 0) // 2
        pl = max((Wp - 1) * 2 + 3 - Ws, 0) // 2
        g.tensor("pool", Hp, Wp, 64)
        g.ops.append(dict(op="maxpool", out="pool", inp="stem", k=3, stride=2, pad_top=pt, pad_left=pl))
        x, cin = "pool", 64
        feats = {}
        for gi, (filters, blocks, stride) in enumerate(zip([64, 128, 256, 512], _MODEL_CONFIG[self.depth],
                                                           [1, 2, 2, 2])):
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
            feats[str(gi + 2)] = Sym(g, x)
        return feats
