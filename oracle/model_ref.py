"""CPU restatement (PyTorch-CPU, float32) of the reference's RetinaNet forward pass.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  PARITY UNPINNED: TensorFlow is absent, so
this follows the reference's Python (cited per function) plus TF 2.8's documented layer
semantics: Conv2D SAME/VALID padding, MaxPool SAME pads
bottom/right, BatchNormalization inference = gamma*(x-mean)/sqrt(var+eps)+beta.

Written independently of retinanet-tensorflow2.x_amd/retinanet/model/graph.py (it walks the
reference's builder order itself), so agreement also checks the product's graph wiring.
`emulate_bf16=True` rounds weights and every layer output to bfloat16 where the Keras
mixed_bfloat16 policy would (layer outputs are bf16, variables are cast at use, the two
prediction convs run in float32: detection_head.py:87), which lets the HIP path be compared
with a tight tolerance.  Also the cpu_baseline leg of bench.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def _r(x, on):
    return x.to(torch.bfloat16).to(torch.float32) if on else x


class RefModel:
    def __init__(self, params, variables, emulate_bf16=False, sync_bn_names=False):
        self.p = params
        self.v = {k: v.detach().to("cpu", torch.float32) for k, v in variables.items()}
        self.bf = emulate_bf16
        self.eps = float(params.architecture.batch_norm.epsilon)
        self.bn_tag = "sync_batch_normalization" if sync_bn_names else "batch_normalization"
        self._i = 0

    # ---- primitives (NCHW internally) ------------------------------------------------------------
    def _conv(self, x, name, stride=1, pad=None, f32=False):
        w = self.v[name + "/kernel"]  # HWIO
        k = w.shape[0]
        if pad is None:
            pad = (k - 1) // 2
        wt = w.permute(3, 2, 0, 1).contiguous()  # OIHW
        # weights are rounded to bf16 on the GPU path for every conv, the fp32-labelled
        # prediction convs included (documented deviation: DESIGN.md "prediction convs")
        wt = _r(wt, self.bf)
        b = self.v.get(name + "/bias")
        return F.conv2d(x, wt, b, stride=stride, padding=pad)

    def _bn(self, x, name):
        g, b = self.v[name + "/gamma"], self.v[name + "/beta"]
        m, var = self.v[name + "/moving_mean"], self.v[name + "/moving_variance"]
        s = g / torch.sqrt(var + self.eps)
        return x * s[None, :, None, None] + (b - m * s)[None, :, None, None]

    def _act(self, x, kind):
        if kind == "relu":
            return F.relu(x)
        if kind == "relu6":
            return F.relu6(x)
        if kind == "swish":
            return x * torch.sigmoid(x)
        return x

    def _next(self):
        i = self._i
        self._i += 1
        c = "conv2d" if i == 0 else f"conv2d_{i}"
        b = self.bn_tag if i == 0 else f"{self.bn_tag}_{i}"
        return c, b

    # ---- resnet.py:194-248, 289-341 -------------------------------------------------------------
    def _bottleneck(self, x, filters, stride, proj):
        sc = x
        if proj:
            c, b = self._next()
            sc = _r(self._bn(self._conv(x, c, stride, pad=0), b), self.bf)
        c, b = self._next()
        y = _r(F.relu(self._bn(self._conv(x, c, 1), b)), self.bf)
        c, b = self._next()
        y = _r(F.relu(self._bn(self._conv(y, c, stride, pad=1), b)), self.bf)
        c, b = self._next()
        y = self._bn(self._conv(y, c, 1), b)
        return _r(F.relu(y + sc), self.bf)

    def backbone(self, images_nhwc):
        self._i = 0
        x = _r(images_nhwc.permute(0, 3, 1, 2).contiguous(), self.bf)
        c, b = self._next()
        x = _r(F.relu(self._bn(self._conv(x, c, 2, pad=3), b)), self.bf)
        # MaxPool 3x3 s2 SAME: pad bottom/right only when the input is even (TF rule)
        H, W = x.shape[2], x.shape[3]
        ph = max((math.ceil(H / 2) - 1) * 2 + 3 - H, 0)
        pw = max((math.ceil(W / 2) - 1) * 2 + 3 - W, 0)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
        x = F.max_pool2d(x, 3, 2)
        feats = {}
        depth = int(self.p.architecture.backbone.depth)
        for gi, (f, n, s) in enumerate(zip([64, 128, 256, 512], _LAYERS[depth], [1, 2, 2, 2])):
            for bi in range(n):
                x = self._bottleneck(x, f, s if bi == 0 else 1, bi == 0)
            feats[str(gi + 2)] = x
        return feats

    # ---- fpn_base.py:54-71, fpn.py:81-107 --------------------------------------------------------
    def fpn(self, feats):
        ff = self.p.architecture.feature_fusion
        lo, hi, bmax = ff.min_level, ff.max_level, ff.backbone_max_level
        act = self.p.architecture.activation.type
        t = self.bn_tag
        out = dict(feats)
        for level in range(bmax + 1, hi + 1):
            x = out[str(level - 1)]
            if level == bmax + 1:
                x = _r(self._bn(self._conv(x, "fpn/backbone_max_level_conv_1x1"), f"fpn/backbone_max_level_{t}"),
                       self.bf)
            out[str(level)] = F.max_pool2d(x, 2)
        for level in range(lo, bmax + 1):
            x = self._conv(out[str(level)], f"fpn/p{level}-in-channel-normalize-conv-1x1")
            out[str(level)] = _r(self._bn(x, f"fpn/p{level}-in-channel-normalize-{t}"), self.bf)
        for level in range(hi, lo, -1):
            up = F.interpolate(out[str(level)], scale_factor=2, mode="nearest")
            out[str(level - 1)] = _r(self._act(out[str(level - 1)] + up, act), self.bf)
        for level in range(lo, hi + 1):
            x = self._conv(out[str(level)], f"fpn/p{level}-out-conv-3x3")
            out[str(level)] = _r(self._bn(x, f"fpn/p{level}-out-{t}"), self.bf)
        return {str(l): out[str(l)] for l in range(lo, hi + 1)}

    # ---- balance_features.py:19-60 ---------------------------------------------------------------
    def balance(self, feats):
        ff = self.p.architecture.feature_fusion
        lo, hi = ff.min_level, ff.max_level
        mid = lo + 1
        resized = []
        for level in range(lo, hi + 1):
            x = feats[str(level)]
            if level > mid:
                x = F.interpolate(x, scale_factor=2 ** (level - mid), mode="nearest")
            elif level < mid:
                x = F.max_pool2d(x, 2 ** (mid - level))
            resized.append(x)
        avg = resized[0]
        for x in resized[1:]:
            avg = avg + x
        avg = _r(avg / float(hi - lo + 1), self.bf)
        out = {}
        for level in range(lo, hi + 1):
            if level > mid:
                a = F.max_pool2d(avg, 2 ** (level - mid))
            elif level < mid:
                a = F.interpolate(avg, scale_factor=2 ** (mid - level), mode="nearest")
            else:
                a = avg
            out[str(level)] = _r(feats[str(level)] + a, self.bf)
        return out

    # ---- detection_head.py:90-104 ----------------------------------------------------------------
    def head(self, feats, name):
        hd = self.p.architecture.head
        act = self.p.architecture.activation.type
        outs = {}
        for level, x in feats.items():
            for i in range(hd.num_convs):
                x = self._conv(x, f"{name}/{name}-{i}-conv2d")
                x = self._bn(x, f"{name}/{name}-{i}-p{level}-{self.bn_tag}")
                x = _r(self._act(x, act), self.bf)
            y = self._conv(x, f"{name}/{name}-prediction-conv2d", f32=True)
            outs[level] = y.permute(0, 2, 3, 1).contiguous()  # NHWC float32
        return outs

    def __call__(self, images_nhwc):
        """images f32[B,H,W,3] -> {'class-predictions': {...}, 'box-predictions': {...}} NHWC f32."""
        with torch.no_grad():
            feats = self.fpn(self.backbone(images_nhwc.to(torch.float32)))
            if self.p.architecture.feature_fusion.use_balanced_features:
                feats = self.balance(feats)
            return {"class-predictions": self.head(feats, "class-head"),
                    "box-predictions": self.head(feats, "box-head"),
                    "_features": {k: v.permute(0, 2, 3, 1).contiguous() for k, v in feats.items()}}
