"""FPN as static graph ops — retinanet/model/neck/fpn_base.py:54-71 (P6 = maxpool2(BN(conv1x1(C5))), P7 =
maxpool2(P6)) and retinanet/model/neck/fpn.py:11-107 (lateral 1x1 + BN, top-down `act(P_{l-1} + up2(P_l))` for
l = max..min+1 — P6 and P5 are refined from above too, P7 gets no activation —, output 3x3 + BN without activation);
FeatureFusion mode 'sum' (model/layers/feature_fusion.py:41-56), NearestUpsampling2D
(model/layers/nearest_upsampling.py:19-21).  With `conv_2d.use_seperable_conv` every conv is a SeparableConv2D
(fpn_base.py:28-39).  Variable names follow the Keras layer names under the `fpn/` scope."""
from __future__ import annotations

from retinanet.model.graph import Sym, _conv_or_sep


class FPN:
    def __init__(self, filters, min_level, max_level, backbone_max_level, fusion_mode="sum", conv_2d_op_params=None,
                 normalization_op_params=None, activation_fn=None, name="fpn", **_):
        if fusion_mode != "sum":
            raise NotImplementedError("fusion_mode other than 'sum' is unused by every shipped config")
        self.filters, self.min_level, self.max_level = int(filters), int(min_level), int(max_level)
        self.backbone_max_level, self.activation_fn, self.name = int(backbone_max_level), activation_fn, name
        self.separable = bool((conv_2d_op_params or {}).get("use_seperable_conv", False))
        self._sync_names = bool((normalization_op_params or {}).get("sync_names", False))

    def __call__(self, features):
        g = next(iter(features.values())).graph
        feats = {lv: t.name for lv, t in features.items()}
        F, lo, hi, bmax, act = self.filters, self.min_level, self.max_level, self.backbone_max_level, self.activation_fn
        separable, pre = self.separable, self.name + "/"
        bn_tag = "sync_batch_normalization" if self._sync_names else "batch_normalization"
        top = feats[str(bmax)]
        g.add_bn_layer(f"{pre}backbone_max_level_{bn_tag}", F)
        _conv_or_sep(g, separable, "fpn_c6pre", top, f"{pre}backbone_max_level_conv_1x1", 1, F, 0.0, "variance_scaling",
                     f"{pre}backbone_max_level_{bn_tag}", None, None if separable else "fpn_1x1")
        prev = "fpn_c6pre"
        for level in range(bmax + 1, hi + 1):
            Hl, Wl = g.tensors[prev][0] // 2, g.tensors[prev][1] // 2
            g.tensor(f"fpn_in{level}", Hl, Wl, F)
            g.ops.append(dict(op="maxpool", out=f"fpn_in{level}", inp=prev, k=2, stride=2, pad_top=0, pad_left=0))
            prev = f"fpn_in{level}"
        for level in range(lo, bmax + 1):
            src = feats[str(level)]
            name = f"{pre}p{level}-in-channel-normalize-conv-1x1"
            bn = f"{pre}p{level}-in-channel-normalize-{bn_tag}"
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
            name = f"{pre}p{level}-out-conv-3x3"
            bn = f"{pre}p{level}-out-{bn_tag}"
            g.add_bn_layer(bn, F)
            src = f"fpn_td{level}" if level != hi else f"fpn_in{hi}"
            _conv_or_sep(g, separable, f"fpn_out{level}", src, name, 3, F, 0.0, "variance_scaling", bn, None, "fpn_out")
        return {str(l): Sym(g, f"fpn_out{l}") for l in levels}
