"""`build_detection_heads(params, min_level, max_level, ...)` — retinanet/model/head/builder.py:7-43: the box head
(4 * num_anchors outputs, zero prediction bias) and the class head (num_anchors * num_classes outputs, prediction
bias -log((1 - 0.01) / 0.01))."""
from __future__ import annotations

import numpy as np

from retinanet.model.head.detection_head import DetectionHead


def build_detection_heads(params, min_level, max_level, conv_2d_op_params=None, normalization_op_params=None,
                          activation_fn=None):
    if activation_fn is None:
        raise ValueError("`activation_fn` cannot be None")
    box_head = DetectionHead(num_convs=params.num_convs, filters=params.filters,
                             output_filters=params.num_anchors * 4, min_level=min_level, max_level=max_level,
                             prediction_bias_initializer="zeros", conv_2d_op_params=conv_2d_op_params,
                             normalization_op_params=normalization_op_params, activation_fn=activation_fn,
                             name="box-head")
    prior_prob_init = -float(np.log((1 - 0.01) / 0.01))
    class_head = DetectionHead(num_convs=params.num_convs, filters=params.filters,
                               output_filters=params.num_anchors * params.num_classes, min_level=min_level,
                               max_level=max_level, prediction_bias_initializer=prior_prob_init,
                               conv_2d_op_params=conv_2d_op_params, normalization_op_params=normalization_op_params,
                               activation_fn=activation_fn, name="class-head")
    return box_head, class_head


def build_auxillary_head(num_convs, filters, num_anchors, min_level, max_level, conv_2d_op_params=None,
                         normalization_op_params=None, activation_fn=None):
    """head/builder.py:46-72.  `use_auxillary_head` is false in every shipped config: the loss has no
    iou-prediction term to train it with (retinanet_loss.py:66-83 reports 0), so the graph does not build it."""
    raise NotImplementedError("auxillary head is disabled in every shipped config")
