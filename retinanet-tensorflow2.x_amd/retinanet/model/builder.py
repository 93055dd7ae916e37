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
import os
import re
from collections import OrderedDict

import numpy as np
import torch

from retinanet.dataloader.anchor_generator import AnchorBoxGenerator
from retinanet.losses import RetinaNetLoss
from retinanet.model.engine import InferenceEngine
from retinanet.model.graph import build_retinanet_graph, init_variables
from retinanet.model.layers import DetectionPostProcess


class LayerView:
    """One entry of `model.layers` as Executor._maybe_freeze_layers / _get_weight_decay_variables walk them
    (executor.py:154-176, 296-327, after `_maybe_flatten_layers`): `.name`, `.weights` (objects with `.name`),
    `.trainable` (setting it False freezes every variable of the layer, Keras semantics)."""

    class _W:
        def __init__(self, name):
            self.name = name

    def __init__(self, model, name, var_names):
        self._model, self.name, self.var_names = model, name, list(var_names)
        self.weights = [LayerView._W(n) for n in self.var_names]

    @property
    def trainable(self):
        return not all(n in self._model._frozen for n in self.var_names)

    @trainable.setter
    def trainable(self, value):
        if value:
            self._model._frozen.difference_update(self.var_names)
        else:
            self._model._frozen.update(self.var_names)
        self._model._train_engines.clear()


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
        self._train_engines = {}
        # rn_launch_opts (dict of fields or _C.LaunchOpts) for the engines this model builds: kernel-family overrides
        # for tests / A/B timing, per model — the library has no process-wide knobs.  Set before the first call.
        self.launch_opts = None
        self._frozen = set()
        self.loaded_extras = {}
        self.input_shape = (None,) + tuple(graph.tensors["images"][:3])

    @property
    def layers(self):
        """Layers at the granularity the reference's executor sees after flattening one level: every ResNet conv /
        BatchNorm layer on its own (the backbone is a nested functional model), every top-level EfficientNet
        sub-layer (stem, blocks_i), and the custom layers `fpn`, `box-head`, `class-head` as ONE layer each."""
        groups = OrderedDict()
        for k in self.variables:
            parts = k.split("/")
            if parts[0] in ("fpn", "box-head", "class-head"):
                key = parts[0]
            elif parts[0].startswith("efficientnet"):
                key = "/".join(parts[:2])
            else:
                key = parts[0]
            groups.setdefault(key, []).append(k)
        return [LayerView(self, name, names) for name, names in groups.items()]

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
        """variable-level freeze (tests, tools); the executor freezes by LAYER through `.layers`"""
        for k in self.variables:
            if regex.search(k):
                self._frozen.add(k)
        self._train_engines.clear()

    @property
    def frozen_variable_names(self):
        return set(self._frozen)

    def _refresh(self):
        for eng in self._engines.values():
            with torch.cuda.device(self.device):
                eng.load_variables(self.variables)

    def inference_engine(self, batch_size, capture_graph=False):
        key = (int(batch_size), bool(capture_graph))
        if key not in self._engines:
            self._engines[key] = InferenceEngine(self.graph, self.variables, batch_size, self.device,
                                                 bn_epsilon=self.params.architecture.batch_norm.epsilon,
                                                 capture_graph=capture_graph,
                                                 f16=(str(self.params.floatx.precision) == "mixed_float16"
                                                      and os.environ.get("RNET_F16", "1") != "0"),
                                                 launch_opts=self.launch_opts)
        return self._engines[key]

    def train_engine(self, batch_size, process_group=None, world_size=None):
        """The TrainEngine that serves `model(images, training=True)` and Executor._train_step for this batch size
        (built once; rebuilt when the set of frozen layers changes)."""
        from retinanet.model.train_engine import TrainEngine
        key = (int(batch_size), world_size)
        if key not in self._train_engines:
            self._train_engines[key] = TrainEngine(self, batch_size, frozen_names=self._frozen,
                                                   process_group=process_group, world_size=world_size,
                                                   launch_opts=self.launch_opts)
        return self._train_engines[key]

    def __call__(self, images, training=False):
        """model/builder.py:94-106: images f32[B,H,W,3] -> {'class-predictions': {'3'..'7': f32[B,s,s,A*K]},
        'box-predictions': {'3'..'7': f32[B,s,s,4A]}}.  training=True runs the training forward (batch-statistics
        BatchNorm on the live layers, frozen layers in inference mode — executor.py:154-176) and leaves the saved
        activations in the engine for `backward`."""
        if training:
            return self.train_engine(images.shape[0]).forward(images)
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
        if device is None:   # one process per GPU: LOCAL_RANK picks it (retinanet/distribute.py)
            import os
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        self._device = torch.device(device)
        self._seed = seed

    def __call__(self):
        params = self.params
        graph = build_retinanet_graph(params)
        variables = init_variables(graph, seed=self._seed, device=self._device)
        loss_fn = RetinaNetLoss(params.architecture.head.num_classes, params.loss)
        model = RetinaNetModel(params, graph, variables, self._device, loss_fn=loss_fn)
        # model/builder.py:108-117: the optimizer is always built and compiled into the model (the export and eval
        # paths read optimizer.iterations / the moving averages)
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
        """model/builder.py:153-190: images -> {'boxes', 'scores', 'classes', 'valid_detections'}.

        The returned dict holds the stage's STATIC output buffers: the next call with the same batch size overwrites
        them (with `capture_graph=True` asynchronously, by one graph replay).  A caller that keeps detections across
        calls — an evaluation loop accumulating results — must `.clone()` them or copy them to the host first."""
        params = self.params
        logging.info("Postprocessing stage config:\n%s", json.dumps(params.inference, indent=4))
        if skip_decoding or skip_nms:
            raise NotImplementedError("skip_decoding / skip_nms only serve the TensorRT export path, "
                                      "which is out of scope (SURVEY §2.1 row 20)")
        ff = params.architecture.feature_fusion
        anchors = AnchorBoxGenerator(*params.input.input_shape, ff.min_level, ff.max_level, params.anchor_params,
                                     device=model.device)
        post = DetectionPostProcess(params, anchors=anchors)

        # capture_graph: batch size -> (engine, HIP graph of forward + post-processing, static outputs, the stage).  A
        # captured graph has the raw addresses of its DetectionPostProcess's boxes / workspace / outputs baked in and that
        # object re-allocates them when the batch size changes, so every captured batch size gets its OWN stage object,
        # kept alive here next to its graph (one shared stage left the first graph replaying into freed memory
        # after a second batch size warmed up).
        graphs = {}

        def inference_model(images, training=False):
            if not capture_graph:
                eng = model.inference_engine(images.shape[0])
                return post(eng(images))
            # `serving_default` as ONE graph launch: the engine's launch list AND the post-processing stage's launches
            # (decode, compaction, per-class NMS, merge) are captured together — at batch 1 the step is ~75 short
            # launches and the host-side dispatch of the last eight was a tenth of the latency.  Every buffer of the
            # stage is static (DetectionPostProcess keeps its boxes / workspace / outputs), so the replay reads and
            # writes the same addresses; the returned dict is overwritten by the next call.
            B = int(images.shape[0])
            st = graphs.get(B)
            if st is None:
                eng = model.inference_engine(B)
                if tuple(images.shape) != tuple(eng.t["images"].shape):
                    raise ValueError(f"expected images of shape {tuple(eng.t['images'].shape)}, got {tuple(images.shape)}")
                post_b = DetectionPostProcess(params, anchors=anchors)
                with torch.cuda.device(model.device):
                    post_b(eng(images))                     # warm-up outside capture: lazy loads, workspace allocation
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        eng._launch_all()
                        out = post_b(eng.outputs)
                st = graphs[B] = (eng, g, out, post_b)
            eng, g, out, _ = st
            if tuple(images.shape) != tuple(eng.t["images"].shape):
                raise ValueError(f"expected images of shape {tuple(eng.t['images'].shape)}, got {tuple(images.shape)}")
            with torch.cuda.device(model.device):
                if images.data_ptr() != eng.t["images"].data_ptr():
                    eng.t["images"].copy_(images, non_blocking=True)
                g.replay()
            return out
        inference_model.post = post
        inference_model.model = model
        return inference_model
