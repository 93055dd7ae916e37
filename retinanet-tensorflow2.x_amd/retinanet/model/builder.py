"""ModelBuilder — the builder surface of the reference (retinanet/model/builder.py:17-190)
on top of the HIP engines.

    model = ModelBuilder(params, run_mode)()            # random-initialised RetinaNet
    preds = model(images, training=False)               # {'class-predictions': {'3'..'7'}, 'box-predictions': ...}
    infer = builder.add_post_processing_stage(model)    # images -> boxes/scores/classes/valid_detections

Differences forced by the platform: tensors are torch tensors on the MI355X; the batch size is
static per engine (the reference does the same for export, builder.py:40-42); `training=True`
is served by `retinanet.model.train_engine` (see DESIGN.md for the rows still in progress).
"""
from __future__ import annotations

import json
import logging
import re
from collections import OrderedDict

import numpy as np
import torch

from retinanet.dataloader.anchor_generator import AnchorBoxGenerator
from retinanet.losses import RetinaNetLoss
from retinanet.model.engine import InferenceEngine
from retinanet.model.graph import build_retinanet_graph, init_variables
from retinanet.model.layers import DetectionPostProcess


class RetinaNetModel:
    name = "retinanet"

    def __init__(self, params, graph, variables, device, loss_fn=None):
        self.params = params
        self.graph = graph
        self.variables = variables  # OrderedDict name -> f32 tensor (conv kernels HWIO)
        self.device = torch.device(device)
        self.loss = loss_fn
        self.optimizer = None
        self._engines = {}
        self._frozen = set()
        self.loaded_extras = {}

    # -- Keras-like surface used by the reference's Executor (executor.py:119,144,244,259,543) --
    @property
    def trainable_variables(self):
        return [v for k, v in self.variables.items()
                if self.graph.var_specs[k].get("trainable", True) and k not in self._frozen]

    @property
    def trainable_variable_names(self):
        return [k for k in self.variables
                if self.graph.var_specs[k].get("trainable", True) and k not in self._frozen]

    def get_weights(self):
        return [v.detach().cpu().numpy() for v in self.variables.values()]

    def set_weights(self, weights):
        if len(weights) != len(self.variables):
            raise ValueError(f"expected {len(self.variables)} arrays, got {len(weights)}")
        for (k, v), w in zip(self.variables.items(), weights):
            v.copy_(torch.as_tensor(w).reshape(v.shape))
        self._refresh()

    def save_weights(self, path, slots=None, extra=None):
        """`<path>.index` + `<path>.data-00000-of-00001` in TensorFlow's checkpoint format (what the reference's
        `model.save_weights`, executor.py:652-654, writes; see retinanet/tf_checkpoint.py), plus the `checkpoint`
        state file `latest_checkpoint` reads.  `slots`: optional {(variable, slot): array} optimizer state
        (TrainEngine.optimizer_slots()).  A path ending in `.safetensors` writes one safetensors file instead."""
        if str(path).endswith(".safetensors"):
            from safetensors.torch import save_file
            save_file({k: v.detach().cpu().contiguous() for k, v in self.variables.items()}, path)
            return
        from retinanet import tf_checkpoint
        arrays = {k: v.detach().cpu().numpy() for k, v in self.variables.items()}
        arrays.update(extra or {})            # e.g. the optimizer's step counter
        tf_checkpoint.save_weights(path, arrays, slots)

    def load_weights(self, path, by_name=True, skip_mismatch=False):
        """Loads a TensorFlow checkpoint prefix (variables matched through the object graph's full names) or a
        `.safetensors` file.  Returns the optimizer slots found in the checkpoint ({} for safetensors)."""
        slots = {}
        if str(path).endswith(".safetensors"):
            from safetensors.torch import load_file
            loaded = load_file(path)
        else:
            from retinanet import tf_checkpoint
            arrays, slots = tf_checkpoint.load_weights(path)
            loaded = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in arrays.items()}
            self.loaded_extras = {k: a for k, a in arrays.items() if k not in self.variables}
        for k, v in self.variables.items():
            if k not in loaded:
                if skip_mismatch:
                    continue
                raise KeyError(f"{k} missing from {path}")
            if tuple(loaded[k].shape) != tuple(v.shape):
                if skip_mismatch:
                    continue
                raise ValueError(f"{k}: shape {tuple(loaded[k].shape)} != {tuple(v.shape)}")
            v.copy_(loaded[k])
        self._refresh()
        return slots

    def summary(self, print_fn=print):
        n = sum(v.numel() for k, v in self.variables.items() if self.graph.var_specs[k].get("trainable", True))
        print_fn(f"retinanet: {len(self.variables)} variables, {n:,} trainable parameters")

    def freeze(self, regex):
        for k in self.variables:
            if regex.search(k):
                self._frozen.add(k)

    def _refresh(self):
        for eng in self._engines.values():
            with torch.cuda.device(self.device):
                eng.load_variables(self.variables)

    def inference_engine(self, batch_size, capture_graph=False):
        key = (int(batch_size), bool(capture_graph))
        if key not in self._engines:
            self._engines[key] = InferenceEngine(self.graph, self.variables, batch_size, self.device,
                                                 bn_epsilon=self.params.architecture.batch_norm.epsilon,
                                                 capture_graph=capture_graph)
        return self._engines[key]

    def __call__(self, images, training=False):
        if training:
            raise NotImplementedError("training forward is served by retinanet.model.train_engine")
        return self.inference_engine(images.shape[0])(images)


class ModelBuilder:
    FREEZE_VARS_REGEX = {
        "backbone": re.compile(r"^(?!((fpn)|(box-head)|(class-head)))"),
        "backbone-bn": re.compile(r"^(?!((fpn)|(box-head)|(class-head))).*(batch_normalization)"),
        "fpn": re.compile(r"^(fpn)"),
        "fpn-bn": re.compile(r"^(fpn).*(batch_normalization)"),
        "head": re.compile(r"^((box-head)|(class-head))(?!.*prediction)"),
        "head-bn": re.compile(r"^((box-head)|(class-head)).*(batch_normalization)"),
        "bn": re.compile(r"(batch_normalization)"),
        "resnet_initial": re.compile(
            r"^(?!((fpn)|((stacked_)?mlaf)|(box-head)|(class-head))).*"
            r"(conv2d(_fixed_padding)?(|_([1-9]|10))|(sync_)?batch_normalization(|_([1-9]|10)))\/"),
    }

    def __init__(self, params, run_mode, device=None, seed=1337):
        self.params = params
        self._run_mode = run_mode
        self._device = torch.device(device if device is not None else "cuda")
        self._seed = seed

    def __call__(self):
        params = self.params
        graph = build_retinanet_graph(params)
        variables = init_variables(graph, seed=self._seed, device=self._device)
        loss_fn = RetinaNetLoss(params.architecture.head.num_classes, params.loss)
        model = RetinaNetModel(params, graph, variables, self._device, loss_fn=loss_fn)
        if "train" in self._run_mode:
            from retinanet.optimizers import build_optimizer
            model.optimizer = build_optimizer(params.training.optimizer, params.training.train_steps,
                                              precision=params.floatx.precision)
        return model

    def prepare_model_for_export(self, model, mode="tf"):
        model.optimizer = None
        skip_decoding = skip_nms = False
        if mode == "tf":
            pass
        elif mode in ("tf_tensorrt", "onnx"):
            if self.params.inference.pre_nms_top_k > 0:
                logging.warning("Forcefully disabling top-k filtering (reference builder.py:134-139)")
                self.params.inference.pre_nms_top_k = -1
        elif mode == "onnx_tensorrt":
            skip_decoding = skip_nms = True
        else:
            raise ValueError("Invalid export model requested!")
        return self.add_post_processing_stage(model, skip_decoding=skip_decoding, skip_nms=skip_nms)

    def add_post_processing_stage(self, model, skip_decoding=False, skip_nms=False, capture_graph=False):
        params = self.params
        logging.info("Postprocessing stage config:\n%s", json.dumps(params.inference, indent=4))
        if skip_decoding or skip_nms:
            raise NotImplementedError("skip_decoding / skip_nms only serve the TensorRT export path, "
                                      "which is out of scope (SURVEY §2.1 row 20)")
        ff = params.architecture.feature_fusion
        anchors = AnchorBoxGenerator(*params.input.input_shape, ff.min_level, ff.max_level, params.anchor_params,
                                     device=model.device)
        post = DetectionPostProcess(params, anchors=anchors)

        def inference_model(images, training=False):
            eng = model.inference_engine(images.shape[0], capture_graph=capture_graph)
            return post(eng(images))
        inference_model.post = post
        inference_model.model = model
        return inference_model
