"""`build_backbone(input_shape, params, normalization_op_params)` — retinanet/model/backbone/builder.py:7-33."""
from __future__ import annotations

from copy import deepcopy

from retinanet.model.backbone.efficientnet import EfficientNet
from retinanet.model.backbone.resnet import ResNet


def build_backbone(input_shape, params, normalization_op_params=None):
    kind = params.type.lower()
    if "resnet" in kind:
        resnet_params = dict(deepcopy(params))
        resnet_params.pop("type")
        return ResNet(input_shape=input_shape, normalization_op_params=normalization_op_params, **resnet_params)
    if "efficientnet" in kind:
        return EfficientNet(input_shape=input_shape, model_name=params.type, checkpoint=params.get("checkpoint", ""),
                            normalization_op_params=normalization_op_params,
                            override_params=params.get("override_params", None))
    if "mobiledet" in kind:
        raise NotImplementedError("MobileDet backbones are out of scope (SURVEY §8: the hot path is ResNet / "
                                  "EfficientNet-B RetinaNet)")
    raise ValueError("{} backbone not implemented".format(params.type))
