"""CPU restatement (PyTorch-CPU, float32) of the reference's RetinaNet forward pass.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).  PARITY UNPINNED: TensorFlow is absent, so
this follows the reference's Python (cited per function) plus TF 2.8's documented layer
semantics: Conv2D SAME/VALID padding, MaxPool SAME pads
bottom/right, BatchNormalization inference = gamma*(x-mean)/sqrt(var+eps)+beta.

Written independently of retinanet-tensorflow2.x_amd/retinanet/model/graph.py (it walks the
reference's builder order itself), so agreement also checks the product's graph wiring.
`emulate_bf16=True` rounds weights and every LAYER output to bfloat16 where the Keras
mixed_bfloat16 policy (__main__.py:76-77) makes it a bf16 tensor: the output of Conv2D /
DepthwiseConv2D (conv + bias: one layer, one rounding), of BatchNormalization, of the residual
`+` (resnet.py:248) / FeatureFusion Add, of drop_connect, of the activation; variables are cast
at use.  The two prediction convs are built with dtype=float32 (detection_head.py:80-88): their
input is cast UP, their kernel and bias stay float32, nothing is rounded.  With the rounding
points shared, the HIP path differs from this file only by fp32 summation order.
Also the cpu_baseline leg of bench.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

_LAYERS = {14: [1, 1, 1, 1], 26: [2, 2, 2, 2], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


class _RoundBF16(torch.autograd.Function):
    """Round to the policy's 16-bit type (bfloat16 under mixed_bfloat16, float16 under mixed_float16) in the forward
    AND the backward pass: what materialising an activation (and, in training, its gradient) as a 16-bit tensor does
    on the GPU path."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def _r(x, on):
    """`on`: False / None = no rounding, True = bfloat16, or the torch dtype itself (torch.float16: mixed_float16)."""
    if not on:
        return x
    dt = torch.bfloat16 if on is True else on
    if x.requires_grad:
        return _RoundBF16.apply(x, dt)
    return x.to(dt).to(x.dtype)


class RefModel:
    def __init__(self, params, variables, emulate_bf16=False, sync_bn_names=False):
        self.p = params
        self.v = {k: v.detach().to("cpu", torch.float32) for k, v in variables.items()}
        # True = the 16-bit type of the config's mixed-precision policy (float16 under mixed_float16, else bfloat16)
        if emulate_bf16 is True and str(getattr(getattr(params, "floatx", None), "precision", "")) == "mixed_float16":
            emulate_bf16 = torch.float16
        self.bf = emulate_bf16
        self.eps = float(params.architecture.batch_norm.epsilon)
        self.bn_tag = "sync_batch_normalization" if sync_bn_names else "batch_normalization"
        self._i = 0

    # ---- primitives (NCHW internally) ------------------------------------------------------------
    def _conv(self, x, name, stride=1, pad=None, f32=False, kernel="/kernel"):
        w = self.v[name + kernel]  # HWIO
        k = w.shape[0]
        if pad is None:
            pad = (k - 1) // 2
        wt = w.permute(3, 2, 0, 1).contiguous()  # OIHW
        b = self.v.get(name + "/bias")
        if f32:   # layer dtype float32 (detection_head.py:80-88): f32 kernel, f32 output, no rounding
            return F.conv2d(x, wt, b, stride=stride, padding=pad)
        # compute dtype bf16: kernel cast at use, fp32 accumulate, + bias, the layer's output is a bf16 tensor
        return _r(F.conv2d(x, _r(wt, self.bf), b, stride=stride, padding=pad), self.bf)

    @staticmethod
    def _same_pad(x, k, s, value=0.0):
        """TF SAME: total = max((ceil(n/s)-1)*s + k - n, 0); the odd element goes bottom/right."""
        H, W = x.shape[2], x.shape[3]
        ph = max((math.ceil(H / s) - 1) * s + k - H, 0)
        pw = max((math.ceil(W / s) - 1) * s + k - W, 0)
        return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=value)

    def _depthwise(self, x, var, stride=1, f32=False):
        """tf.keras DepthwiseConv2D / the depthwise half of SeparableConv2D, padding='same', no bias."""
        w = self.v[var]  # [k,k,C,1]
        k, C = w.shape[0], w.shape[2]
        wt = w.permute(2, 3, 0, 1).contiguous()  # [C,1,k,k]
        if f32:   # inside a dtype=float32 SeparableConv2D (the prediction layers): nothing is rounded
            return F.conv2d(self._same_pad(x, k, stride), wt, None, stride=stride, groups=C)
        return _r(F.conv2d(self._same_pad(x, k, stride), _r(wt, self.bf), None, stride=stride, groups=C), self.bf)

    def _cs(self, x, name, f32=False):
        """Conv2D or SeparableConv2D, stride 1, SAME, bias (fpn_base.py:28-39, detection_head.py:37-50)."""
        if not self.p.architecture.conv_2d.use_seperable_conv:
            return self._conv(x, name, f32=f32)
        y = self._depthwise(x, name + "/depthwise_kernel", f32=f32)
        return self._conv(y, name, pad=0, f32=f32, kernel="/pointwise_kernel")

    def _bn(self, x, name):
        g, b = self.v[name + "/gamma"], self.v[name + "/beta"]
        m, var = self.v[name + "/moving_mean"], self.v[name + "/moving_variance"]
        s = g / torch.sqrt(var + self.eps)
        return _r(x * s[None, :, None, None] + (b - m * s)[None, :, None, None], self.bf)

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
        return F.relu(_r(y + sc, self.bf))

    # ---- efficientnet.py:222-265 (SE), :291-482 (MBConvBlock), :566-586 (Stem), :783-855 -------------
    def _efficientnet(self, images_nhwc):
        name = self.p.architecture.backbone.type
        width, depth = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1), "efficientnet-b2": (1.1, 1.2),
                        "efficientnet-b3": (1.2, 1.4), "efficientnet-b4": (1.4, 1.8)}[name]

        def rf(f):  # round_filters :196-211
            f *= width
            n = max(8, int(f + 4) // 8 * 8)
            return int(n + 8 if n < 0.9 * f else n)
        swish = lambda t: t * torch.sigmoid(t)   # on the bf16 tensor the layer in front of it returned
        bnb = "tpu_batch_normalization"
        x = _r(images_nhwc.permute(0, 3, 1, 2).contiguous(), self.bf)
        x = self._conv(self._same_pad(x, 3, 2), name + "/stem/conv2d", 2, pad=0)
        x = _r(swish(self._bn(x, f"{name}/stem/{self.bn_tag}")), self.bf)
        stages = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
                  (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
        blocks = []
        for rep, k, s, e, ci, co in stages:
            ci, co = rf(ci), rf(co)
            for r in range(int(math.ceil(depth * rep))):
                blocks.append((k, s if r == 0 else 1, e, ci if r == 0 else co, co))
        red, feats = 0, {}
        for i, (k, s, e, ci, co) in enumerate(blocks):
            sc = f"{name}/blocks_{i}/"
            inp, nb, nc = x, 0, 0
            bn = lambda j: sc + (bnb if j == 0 else f"{bnb}_{j}")
            cv = lambda j: sc + ("conv2d" if j == 0 else f"conv2d_{j}")
            if e != 1:
                x = _r(swish(self._bn(self._conv(x, cv(nc)), bn(nb))), self.bf)
                nb, nc = nb + 1, nc + 1
            x = _r(swish(self._bn(self._depthwise(x, sc + "depthwise_conv2d/depthwise_kernel", s), bn(nb))), self.bf)
            nb += 1
            # SE.call :252-265; under the mixed policy each op's output is a 16-bit tensor
            se = _r(x.mean(dim=(2, 3), keepdim=True), self.bf)
            se = _r(swish(_r(self._conv(se, sc + "se/conv2d"), self.bf)), self.bf)
            se = _r(self._conv(se, sc + "se/conv2d_1"), self.bf)
            x = _r(_r(torch.sigmoid(se), self.bf) * x, self.bf)
            x = self._bn(self._conv(x, cv(nc)), bn(nb))
            if s == 1 and ci == co:
                m = getattr(self, "drop_connect_factors", {}).get(i)   # training: per-image 0 or 1/survival_prob
                if m is not None:
                    x = _r(x * m.to(x.dtype)[:, None, None, None], self.bf)
                x = _r(x + inp, self.bf)
            if i == len(blocks) - 1 or blocks[i + 1][1] > 1:
                red += 1
                feats[str(red)] = x
        return {str(l): feats[str(l)] for l in range(2, 6)}

    def backbone(self, images_nhwc):
        if self.p.architecture.backbone.type.startswith("efficientnet"):
            return self._efficientnet(images_nhwc)
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
                x = _r(self._bn(self._cs(x, "fpn/backbone_max_level_conv_1x1"), f"fpn/backbone_max_level_{t}"),
                       self.bf)
            out[str(level)] = F.max_pool2d(x, 2)
        for level in range(lo, bmax + 1):
            x = self._cs(out[str(level)], f"fpn/p{level}-in-channel-normalize-conv-1x1")
            out[str(level)] = _r(self._bn(x, f"fpn/p{level}-in-channel-normalize-{t}"), self.bf)
        for level in range(hi, lo, -1):
            up = F.interpolate(out[str(level)], scale_factor=2, mode="nearest")
            out[str(level - 1)] = _r(self._act(out[str(level - 1)] + up, act), self.bf)
        for level in range(lo, hi + 1):
            x = self._cs(out[str(level)], f"fpn/p{level}-out-conv-3x3")
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
                x = self._cs(x, f"{name}/{name}-{i}-conv2d")
                x = self._bn(x, f"{name}/{name}-{i}-p{level}-{self.bn_tag}")
                x = _r(self._act(x, act), self.bf)
            y = self._cs(x, f"{name}/{name}-prediction-conv2d", f32=True)
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


# ==============================================================================================
# Training-step restatement (Executor._train_step, retinanet/executor.py:409-441) with autograd.
class RefTrainer(RefModel):
    """float64 CPU restatement of one training step: training-mode BatchNorm (batch statistics,
    frozen layers in inference mode — executor.py:154-176), RetinaNetLoss
    (losses/retinanet_loss.py:37-83), l2 weight decay on conv kernels of trainable layers
    (executor.py:296-327), per-tensor + global clipping (executor.py:401-407), Keras SGD momentum
    and the tfa moving average (optimizers/builder.py:45-54)."""

    def __init__(self, params, variables, frozen_names=(), emulate_bf16=False, dtype=torch.float64):
        super().__init__(params, variables, emulate_bf16=emulate_bf16)
        self.dtype = dtype
        self.v = {k: v.detach().to("cpu", dtype).clone() for k, v in variables.items()}
        self.frozen = set(frozen_names)
        self.leaf = {}
        for k, t in self.v.items():
            if k.endswith(("/moving_mean", "/moving_variance")) or k in self.frozen:
                continue
            t.requires_grad_(True)
            self.leaf[k] = t
        self.momentum = float(params.architecture.batch_norm.momentum)
        self.new_stats = {}

    def _bn(self, x, name):
        g, b = self.v[name + "/gamma"], self.v[name + "/beta"]
        if (name + "/gamma") in self.frozen:  # layer.trainable=False -> inference mode
            return super()._bn(x, name)
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() / x.shape[1]
        self.new_stats[name + "/moving_mean"] = (self.v[name + "/moving_mean"] * self.momentum
                                                 + mean.detach() * (1 - self.momentum))
        self.new_stats[name + "/moving_variance"] = (self.v[name + "/moving_variance"] * self.momentum
                                                     + var.detach() * n / (n - 1) * (1 - self.momentum))
        xh = (x - mean[None, :, None, None]) / torch.sqrt(var + self.eps)[None, :, None, None]
        return _r(xh * g[None, :, None, None] + b[None, :, None, None], self.bf)

    def forward_train(self, images_nhwc):
        feats = self.fpn(self.backbone(images_nhwc.to(self.dtype)))
        if self.p.architecture.feature_fusion.use_balanced_features:
            feats = self.balance(feats)
        return {"class-predictions": self.head(feats, "class-head"), "box-predictions": self.head(feats, "box-head")}

    def loss(self, preds, cls_t, box_t, num_pos, replicas=1):
        lp = self.p.loss
        K = self.p.architecture.head.num_classes
        B = cls_t.shape[0]
        logits = torch.cat([preds["class-predictions"][l].reshape(B, -1, K) for l in "34567"], dim=1)
        boxes = torch.cat([preds["box-predictions"][l].reshape(B, -1, 4) for l in "34567"], dim=1)
        ct = torch.as_tensor(cls_t, dtype=self.dtype)
        bt = torch.as_tensor(box_t, dtype=self.dtype)
        normalizer = float(num_pos) + 1.0
        y = (torch.arange(K)[None, None, :] == ct.long()[..., None]).to(self.dtype)
        a, gma, ls = lp.focal_loss.alpha, lp.focal_loss.gamma, lp.focal_loss.label_smoothing
        ys = y * (1 - ls) + 0.5 * ls
        ce = torch.clamp(logits, min=0) - logits * ys + torch.log1p(torch.exp(-logits.abs()))
        p = torch.sigmoid(logits)
        at = torch.where(y == 1, torch.tensor(a, dtype=self.dtype), torch.tensor(1 - a, dtype=self.dtype))
        pt = torch.where(y == 1, p, 1 - p)
        fl = at * (1 - pt).pow(gma) * ce * (ct != -2.0).to(self.dtype)[..., None]
        class_loss = fl.sum() / normalizer
        e = boxes - bt
        d = lp.smooth_l1_loss.delta
        hub = torch.where(e.abs() <= d, 0.5 * e * e, d * e.abs() - 0.5 * d * d) * (bt != 0).to(self.dtype)
        box_loss = hub.sum() / 4.0 / normalizer
        weighted = lp.box_loss_weight * box_loss + lp.class_loss_weight * class_loss
        return {"box-loss": box_loss, "class-loss": class_loss, "weighted-loss": weighted}

    def weight_decay(self):
        alpha = self.p.training.weight_decay_alpha
        tot = 0.0
        for k, t in self.leaf.items():
            if "kernel" in k.rsplit("/", 1)[-1]:   # kernel / depthwise_kernel / pointwise_kernel (executor.py:308-327)
                tot = tot + alpha * 0.5 * (t * t).sum()   # tf.nn.l2_loss = sum(w^2)/2
        return tot

    def step(self, images, cls_t, box_t, num_pos, lr, momentum_state=None, ema_state=None, ema_decay=None,
             replicas=1):
        preds = self.forward_train(images)
        losses = self.loss(preds, cls_t, box_t, num_pos, replicas)
        total = losses["weighted-loss"]
        if self.p.training.use_weight_decay:
            total = total + self.weight_decay()
        (total / replicas).backward()
        clip = float(self.p.training.optimizer.clipnorm)
        grads = {k: t.grad.clone() for k, t in self.leaf.items()}
        raw = {k: g.clone() for k, g in grads.items()}
        for k in grads:  # tf.clip_by_norm per tensor, then tf.clip_by_global_norm (executor.py:401-407)
            grads[k] = grads[k] * (clip / max(grads[k].norm().item(), clip))
        gn = math.sqrt(sum(g.norm().item() ** 2 for g in grads.values()))
        for k in grads:
            grads[k] = grads[k] * (clip / max(gn, clip))
        mom = float(self.p.training.optimizer.momentum)
        new_w, new_v, new_e = {}, {}, {}
        for k, t in self.leaf.items():
            v0 = momentum_state[k] if momentum_state else torch.zeros_like(t)
            v1 = mom * v0 - lr * grads[k] * replicas   # all-reduce SUM over identical replicas
            new_v[k] = v1
            new_w[k] = t.detach() + v1
            if ema_state is not None:
                new_e[k] = ema_state[k] - (1 - ema_decay) * (ema_state[k] - new_w[k])
        return {"losses": {k: float(v.detach()) for k, v in losses.items()}, "raw_grads": raw, "clipped_grads": grads,
                "grad_norm": min(gn, clip), "weights": new_w, "momentum": new_v, "ema": new_e,
                "moving_stats": dict(self.new_stats), "preds": preds}
