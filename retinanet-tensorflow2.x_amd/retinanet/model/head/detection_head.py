"""DetectionHead as static graph ops — retinanet/model/head/detection_head.py:8-104: `num_convs` 3x3 convs + bias
whose kernels are SHARED across the pyramid levels (:56-66), one BatchNorm per (conv, level) (:68-74, :99), the
activation, then the prediction conv 3x3 + bias built with dtype=float32 (:80-88: its input is cast up and its
kernel stays float32; here the f32 kernel is carried as split-bf16 planes, rn_conv_segment.w_terms).  The launches
of both heads' i-th tower conv over all levels form ONE grouped launch (`tower{i}`: up to 10 segments), so the
small P5-P7 problems ride along with P3 instead of under-filling the chip."""
from __future__ import annotations

from retinanet.model.graph import Sym, _conv_or_sep


class DetectionHead:
    def __init__(self, num_convs, filters, output_filters, min_level, max_level, prediction_bias_initializer="zeros",
                 conv_2d_op_params=None, normalization_op_params=None, activation_fn=None, name="detection-head", **_):
        self.num_convs, self.filters, self.output_filters = int(num_convs), int(filters), int(output_filters)
        self.min_level, self.max_level, self.activation_fn, self.name = int(min_level), int(max_level), activation_fn, name
        self.prediction_bias = 0.0 if prediction_bias_initializer == "zeros" else float(prediction_bias_initializer)
        self.separable = bool((conv_2d_op_params or {}).get("use_seperable_conv", False))
        self._sync_names = bool((normalization_op_params or {}).get("sync_names", False))
        # detection_head.py:40-43: RandomNormal(stddev=0.01) for Conv2D, the Keras default for SeparableConv2D
        self.kernel_init = "variance_scaling" if self.separable else "normal_0.01"

    def __call__(self, features):
        g = next(iter(features.values())).graph
        head, act, sep = self.name, self.activation_fn, self.separable
        bn_tag = "sync_batch_normalization" if self._sync_names else "batch_normalization"
        levels = list(range(self.min_level, self.max_level + 1))
        key = {"box-head": "box", "class-head": "class"}.get(head, head)
        for i in range(self.num_convs):
            for level in levels:
                g.add_bn_layer(f"{head}/{head}-{i}-p{level}-{bn_tag}", self.filters)
        outs = {}
        for i in range(self.num_convs):
            name = f"{head}/{head}-{i}-conv2d"
            for j, level in enumerate(levels):
                src = features[str(level)].name if i == 0 else f"{head}_t{i - 1}_p{level}"
                _conv_or_sep(g, sep, f"{head}_t{i}_p{level}", src, name, 3, self.filters, 0.0, self.kernel_init,
                             f"{head}/{head}-{i}-p{level}-{bn_tag}", act, f"tower{i}", define=j == 0)
        name = f"{head}/{head}-prediction-conv2d"
        for j, level in enumerate(levels):
            src = f"{head}_t{self.num_convs - 1}_p{level}" if self.num_convs else features[str(level)].name
            _conv_or_sep(g, sep, f"{head}_pred_p{level}", src, name, 3, self.output_filters, self.prediction_bias,
                         self.kernel_init, None, None, f"pred_{key}", out_dtype="f32", define=j == 0)
            outs[str(level)] = Sym(g, f"{head}_pred_p{level}")
        return outs
