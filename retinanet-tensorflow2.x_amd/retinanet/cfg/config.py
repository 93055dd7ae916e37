"""JSON config -> attribute dictionary, same schema as the reference's configs/*.json.

Mirrors `retinanet.cfg.Config(path).params` (reference retinanet/cfg/config.py:8-21, which
wraps json.load in an EasyDict).  `default_params()` builds the ResNet50-640 schema in code so
benchmarks and tests do not need a JSON file on disk.
"""
from __future__ import annotations

import copy
import json


class AttrDict(dict):
    """dict with attribute access, recursively (the subset of EasyDict the reference uses)."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __delattr__(self, k):
        del self[k]

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        def un(v):
            if isinstance(v, dict):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return [un(x) for x in v]
            return v
        return un(self)


class Config:
    def __init__(self, path):
        self.path = path
        with open(path, "r") as fp:
            self._params = AttrDict(json.load(fp))

    @property
    def params(self):
        return self._params


def default_params(input_size=640, batch_train=256, batch_val=8, activation="relu", balanced=True,
                   precision="mixed_bfloat16", inference_batch=1, nms_mode="PerClassHardNMS",
                   strategy="gpu", freeze=("resnet_initial",)):
    """The ResNet50 RetinaNet schema shared by every shipped ResNet config."""
    return AttrDict({
        "experiment": {"name": f"mscoco-retinanet-resnet50-{input_size}x{input_size}", "run_mode": "train",
                       "model_dir": "model_files", "tensorboard_dir": "tensorboard"},
        "input": {"input_shape": [input_size, input_size], "channels": 3},
        "floatx": {"precision": precision},
        "architecture": {
            "conv_2d": {"use_seperable_conv": False, "use_bias_before_bn": False},
            "batch_norm": {"use_sync": True, "momentum": 0.99, "epsilon": 0.001},
            "activation": {"type": activation},
            "backbone": {"type": "resnet", "depth": 50, "checkpoint": ""},
            "feature_fusion": {"type": "fpn", "use_balanced_features": balanced, "fusion_mode": "sum",
                               "filters": 256, "min_level": 3, "max_level": 7, "backbone_max_level": 5},
            "head": {"num_convs": 4, "filters": 256, "num_classes": 80, "num_anchors": 9},
            "auxillary_head": {"use_auxillary_head": False, "num_convs": 2, "filters": 256},
        },
        "loss": {"focal_loss": {"alpha": 0.25, "gamma": 1.5, "label_smoothing": 0.0},
                 "smooth_l1_loss": {"delta": 0.1},
                 "normalizer": {"use_moving_average": False, "momentum": 0.99},
                 "class_loss_weight": 1.0, "box_loss_weight": 50.0, "auxillary_loss_weight": 0.0},
        "training": {
            "use_weight_decay": True, "weight_decay_alpha": 0.0001,
            "batch_size": {"train": batch_train, "val": batch_val},
            "strategy": {"type": strategy, "name": ""},
            "restore_checkpoint": False, "freeze_variables": list(freeze),
            "train_steps": 16875, "validation_samples": 4952, "validation_freq": -1,
            "annotation_file_path": "./instances_val2017.json", "remap_class_ids": True,
            "steps_per_execution": 128, "save_every": 2560,
            "recovery": {"use_inflection_detector": False, "metric_key": "l2-regularization",
                         "threshold": 0.05, "max_trials": 10},
            "optimizer": {"name": "sgd", "momentum": 0.9, "nesterov": False, "clipnorm": 10.0,
                          "use_moving_average": True, "moving_average_decay": 0.9998,
                          "lr_params": {"schedule_type": "cosine_decay", "initial_learning_rate": 0.32,
                                        "warmup_learning_rate": 0.008, "alpha": 0.0001,
                                        "warmup_steps": 500}},
        },
        "fine_tuning": {"fine_tune": False, "pretrained_checkpoint": ""},
        "anchor_params": {"areas": [1024.0, 4096.0, 16384.0, 65536.0, 262144.0],
                          "aspect_ratios": [0.5, 1.0, 2.0],
                          "scales": [1, 1.2599210498948732, 1.5874010519681994]},
        "encoder_params": {"match_iou": 0.5, "ignore_iou": 0.5, "box_variance": [0.1, 0.1, 0.2, 0.2],
                           "scale_box_targets": False},
        "dataloader_params": {"tfrecords": {"train": "coco_remapped_tfrecords/train*",
                                            "val": "coco_remapped_tfrecords/val*"},
                              "augmentations": {"use_augmentation": True, "horizontal_flip": True,
                                                "scale_jitter": {"min_scale": 0.1, "max_scale": 2.0}},
                              "preprocessing": {"mean": [0.485, 0.456, 0.406],
                                                "stddev": [0.229, 0.224, 0.225], "pixel_scale": 255.0},
                              "shuffle_buffer_size": 1024},
        "inference": {"batch_size": inference_batch, "mode": nms_mode, "iou_threshold": 0.5,
                      "score_threshold": 0.05, "soft_nms_sigma": 0.5, "pre_nms_top_k": 5000,
                      "filter_per_class": True, "max_detections": 100},
    })


def efficientnet_params(model_name="efficientnet-b3", input_size=640, filters=160, num_convs=4, activation="relu",
                        nms_mode="PerClassSoftNMS", precision="mixed_float16", **kw):
    """The EfficientNet + separable-conv FPN/head schema of
    configs/v3-32/mscoco-retinanet-efficientnet-b3-896x896-30x-256.json, at BASELINE.json config 4's
    overrides by default (640x640, soft-NMS, mixed_float16 - computed in bf16 here)."""
    p = default_params(input_size=input_size, activation=activation, balanced=False, precision=precision,
                       nms_mode=nms_mode, freeze=(), **kw)
    p.experiment.name = f"mscoco-retinanet-{model_name}-{input_size}x{input_size}"
    p.architecture.conv_2d.use_seperable_conv = True
    p.architecture.backbone = AttrDict({"type": model_name, "checkpoint": ""})
    p.architecture.feature_fusion.filters = filters
    p.architecture.head.filters = filters
    p.architecture.head.num_convs = num_convs
    return p
