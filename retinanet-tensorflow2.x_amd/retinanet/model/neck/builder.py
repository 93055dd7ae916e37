"""`build_neck(params, conv_2d_op_params, normalization_op_params, activation_fn)` — retinanet/model/neck/builder.py:8-56."""
from __future__ import annotations

from retinanet.model.neck.fpn import FPN


def build_neck(params, conv_2d_op_params=None, normalization_op_params=None, activation_fn=None):
    if activation_fn is None:
        raise ValueError("`activation_fn` cannot be None")
    if params.type == "fpn":
        return FPN(filters=params.filters, min_level=params.min_level, max_level=params.max_level,
                   backbone_max_level=params.backbone_max_level, fusion_mode=params.fusion_mode,
                   conv_2d_op_params=conv_2d_op_params, normalization_op_params=normalization_op_params,
                   activation_fn=activation_fn, name="fpn")
    if params.type in ("multi_level_attention", "stacked_multi_level_attention"):
        raise NotImplementedError("the MLAF necks are out of scope: no shipped config uses them (SURVEY Appendix D)")
    raise ValueError("{} FPN not implemented".format(params.type))
