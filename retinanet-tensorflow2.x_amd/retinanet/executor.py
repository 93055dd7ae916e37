"""Executor — the reference's train / eval driver (retinanet/executor.py:16-778) over the HIP engines.

Same constructor, same `run()` dispatch, same step / checkpoint / evaluation semantics:
  * `_train_step` (executor.py:409-441) = `TrainEngine.train_step`: forward with batch-statistics BatchNorm, RetinaNetLoss,
    l2 weight decay on the kernels of trainable layers, per-tensor + global clipping of the LOCAL gradients, all-reduce SUM,
    SGD momentum + moving average, the loss dict with `total-loss`, `l2-regularization`, `gradient-norm`,
    `num-anchors-matched`;
  * `distributed_train_step(iterator, num_steps)` (:443-453) runs `steps_per_execution` steps and returns the LAST step's
    loss dict averaged over the replicas;
  * `_maybe_freeze_layers` (:154-176) freezes by LAYER through `model.layers` (a custom layer like `fpn` freezes as a whole);
  * checkpoints: `weights_step_<n>` every `save_every` steps and `final_weights_step_<n>` at the end, in TensorFlow's
    checkpoint format (`retinanet.tf_checkpoint`), restored from `tf.train.latest_checkpoint`'s state file (:221-257);
  * `evaluate()` (:477-552): moving-average weights in, `_eval_step` over `val_steps` batches, COCOEvaluator, weights back.
What the MI355X build does differently, by design: one PROCESS per GPU (`strategy` is `retinanet.distribute.Strategy`, not a
tf.distribute strategy), so every rank owns its input pipeline shard; scalars go to `<tensorboard_dir>/<name>/{train,eval}/
scalars.jsonl` (TensorBoard event files, the TF profiler hooks, the graph trace and the Discord hook are the reference's
control plane — out of scope)."""
from __future__ import annotations

import json
import logging
import os
from time import sleep, time

import numpy as np
import torch

from retinanet import tf_checkpoint
from retinanet.eval import COCOEvaluator


class AverageMeter:
    """retinanet/utils.py:7-34"""

    def __init__(self, name=None, momentum=0.997):
        if momentum >= 1 or momentum <= 0:
            raise AssertionError("`momentum` should be a non zero float less than 1")
        self.name, self.momentum, self._averaged_value, self._count = name, momentum, None, 0

    def accumulate(self, x):
        if self._count < 10:
            self._averaged_value = x
        else:
            self._averaged_value = self._averaged_value * self.momentum + (1 - self.momentum) * x
        self._count += 1

    @property
    def averaged_value(self):
        return self._averaged_value


def format_eta(secs):
    """retinanet/utils.py:37-42"""
    eta = []
    for interval, unit in zip([3600, 60, 1], ["h", "m", "s"]):
        eta += ["{:02}{}".format(int(secs // interval), unit)]
        secs %= interval
    return " ".join(eta)


class InflectionDetector:
    """retinanet/loss_diagnostics.py:4-37: flags a jump in the second derivative of a logged metric"""

    def __init__(self, name, threshold, skip_steps=45):
        if skip_steps < 2:
            raise ValueError("`skip_steps` should be greater than 2")
        self.name, self.threshold, self._skip_steps, self._data = name, threshold, skip_steps, []

    def is_value_anomalous(self, value):
        self._data += [value]
        if len(self._data) > self._skip_steps:
            grads = np.gradient(np.gradient(self._data))
            diffs = np.round(np.abs(np.diff(grads)), 3)
            return bool(diffs[-2] > self.threshold)
        return False

    def reset(self):
        self._data = []

    @property
    def data(self):
        return self._data


class _RestoreStatus:
    def __init__(self, missing):
        self._missing = list(missing)

    def assert_consumed(self):
        if self._missing:
            raise AssertionError(f"checkpoint did not provide {self._missing[:5]} (+{max(0, len(self._missing) - 5)} more)")
        return self


class _ScalarWriter:
    def __init__(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        self._fp = open(path, "a")

    def scalars(self, step, values):
        self._fp.write(json.dumps({"step": int(step), **{k: float(v) for k, v in values.items()}}) + "\n")
        self._fp.flush()


def _f(v):
    return float(v.item()) if hasattr(v, "item") else float(v)


class Executor:
    _RUN_MODES = ["train", "val", "train_val", "continuous_eval", "export"]

    def __init__(self, params, strategy, run_mode, model_builder, train_input_fn, val_input_fn=None,
                 is_multi_host=False, enable_weights_info=False, resume_from=None):
        self.params = params
        self.distribute_strategy = strategy
        self.run_mode = run_mode
        self.model_builder = model_builder
        self.restore_checkpoint = params.training.restore_checkpoint
        self.train_input_fn = train_input_fn
        self.val_input_fn = val_input_fn
        self.is_multi_host = is_multi_host
        self.enable_weights_info = enable_weights_info
        self.resume_from = resume_from
        self.train_steps = params.training.train_steps
        self.validation_samples = params.training.validation_samples
        self.val_freq = params.training.validation_freq
        self.steps_per_execution = params.training.steps_per_execution
        self.batch_size = params.training.batch_size
        self.model_dir = os.path.join(params.experiment.model_dir, params.experiment.name)
        self.save_every = params.training.save_every
        self.summary_dir = params.experiment.tensorboard_dir
        self.name = params.experiment.name
        self.val_steps = self.validation_samples // params.training.batch_size["val"]
        self.num_replicas = self.distribute_strategy.num_replicas_in_sync
        self.restore_status = None
        self._clip_gradients = True
        self.use_float16 = False
        self._save_during_training = params.training.save_every > 0
        self._run_evaluation_at_end = params.training.validation_freq < 1
        self._summary_writers = {}
        self._current_trial = 1
        self._engine = None
        self._loaded_slots = {}
        if params.training.recovery.use_inflection_detector:
            self._inflection_detector = InflectionDetector(name=params.training.recovery.metric_key,
                                                           threshold=params.training.recovery.threshold)
            self._max_trials = params.training.recovery.max_trials
        else:
            self._max_trials = 1
        if self.run_mode not in Executor._RUN_MODES:
            raise AssertionError("Invalid run mode, aborting!\n Supported run models {}".format(Executor._RUN_MODES))
        self._setup()

    # ---- public entry (executor.py:94-102) -------------------------------------------------------------------
    def run(self):
        if "train" in self.run_mode:
            self.train()
        elif self.run_mode == "val":
            return self.evaluate()
        elif self.run_mode == "continuous_eval":
            self.continuous_evaluate()

    @property
    def _is_chief(self):
        return getattr(self.distribute_strategy, "rank", 0) == 0

    # ---- set-up (executor.py:104-152, 263-274) -----------------------------------------------------------------
    def _setup(self):
        os.makedirs(self.model_dir, exist_ok=True)
        if self.run_mode == "train" and self._is_chief:
            self.dump_config()
        self._setup_model()
        self._setup_dataset()
        if self.restore_checkpoint or self.run_mode == "export":
            self._restore_checkpoint()

    def _setup_model(self):
        logging.info("Setting up model for %s", self.run_mode)
        self._model = self.model_builder()
        self.optimizer = self._model.optimizer
        opt_cfg = self.params.training.optimizer
        if "global_clipnorm" in opt_cfg or "clipnorm" in opt_cfg:
            self._clip_gradients = True
            logging.warning("Training with `clip_gradients=True`")
        if "train" in self.run_mode and self.optimizer.clipnorm is None:
            # executor.py:432-434 reads optimizer.clipnorm unconditionally — in the train step only: val /
            # continuous_eval / export never clip a gradient
            raise AssertionError("`training.optimizer.clipnorm` is required: gradients are always clipped")
        if self.params.fine_tuning.fine_tune:
            logging.info("Loading pretrained weights for fine-tuning from %s",
                         self.params.fine_tuning.pretrained_checkpoint)
            self._model.load_weights(self.params.fine_tuning.pretrained_checkpoint, skip_mismatch=True, by_name=True)
        logging.info("Trainable variables: %d", len(self._model.trainable_variables))
        if self._maybe_freeze_layers():
            logging.info("Trainable variables after freezing: %d", len(self._model.trainable_variables))
        self.use_float16 = bool(self.optimizer.dynamic_loss_scale)
        if "val" in self.run_mode or self.run_mode == "continuous_eval":
            self._eval_model = self.model_builder.add_post_processing_stage(self._model)
        self._model.summary(print_fn=logging.debug)
        logging.info("Total trainable parameters: {:,}".format(sum(int(v.numel()) for v in self._model.trainable_variables)))
        self._weight_decay_vars = self._get_weight_decay_variables()
        logging.info("Initial weight decay loss: {:.4f}".format(self.weight_decay()))
        if "train" in self.run_mode:
            per_replica = self.batch_size["train"] // self.num_replicas
            if per_replica * self.num_replicas != self.batch_size["train"]:
                raise ValueError(f"`training.batch_size.train` {self.batch_size['train']} is not divisible by "
                                 f"{self.num_replicas} replicas")
            self._engine = self._model.train_engine(per_replica, process_group=getattr(self.distribute_strategy, "group", None),
                                                    world_size=self.num_replicas)
            if self.num_replicas > 1:     # SyncBN / normaliser messages through rn_comm when it validates on every rank
                from retinanet import comm
                comm.maybe_enable_native(self._engine)

    def _maybe_freeze_layers(self):
        patterns = self.params.training.freeze_variables
        if not patterns:
            return False
        for pattern in patterns:
            regex = self.model_builder.FREEZE_VARS_REGEX[pattern]
            logging.warning("Freezing layers with variables that match pattern: %s", regex.pattern)
            for layer in self._model.layers:
                for weight in layer.weights:
                    if regex.search(weight.name) and layer.trainable:
                        layer.trainable = False
                        logging.debug("Freezing layer: %s", layer.name)
        return True

    def _input_context(self):
        """one process per GPU: every rank is an input pipeline of its own (rank-th shard, per-replica batch)"""
        if self.num_replicas <= 1:
            return None
        from retinanet.dataloader.input_pipeline import InputContext
        return InputContext(self.num_replicas, getattr(self.distribute_strategy, "rank", 0), self.num_replicas)

    def _setup_dataset(self):
        if ("val" in self.run_mode or self.run_mode == "continuous_eval") and self.val_input_fn is not None:
            logging.info("Setting up val dataset")
            self._val_dataset = lambda: self.val_input_fn(self._input_context())
        if "train" in self.run_mode:
            logging.info("Setting up train dataset")
            self._train_dataset = lambda: self.train_input_fn(self._input_context())

    def _setup_summary_writers(self):
        if not self._is_chief:
            return
        for key, sub in (("train", "train"), ("eval", "eval")):
            if (key == "train" and "train" in self.run_mode) or (key == "eval" and "val" in self.run_mode) or \
                    (key == "eval" and self.run_mode == "continuous_eval"):
                self._summary_writers.setdefault(key, _ScalarWriter(os.path.join(self.summary_dir, self.name, sub,
                                                                                 "scalars.jsonl")))

    # ---- checkpoints (executor.py:221-257) ---------------------------------------------------------------------
    def _restore_checkpoint(self, checkpoint=None):
        if checkpoint is not None:
            latest = checkpoint
        elif self.resume_from is not None:
            latest = os.path.join(self.model_dir, self.resume_from)
        else:
            logging.info("Looking for existing checkpoints in %s", self.model_dir)
            latest = tf_checkpoint.latest_checkpoint(self.model_dir)
        if latest is not None:
            logging.info("Found existing checkpoint %s, restoring model and optimizer state from checkpoint", latest)
            before = set(self._model.variables)
            if self._engine is not None:
                self._engine.restore_checkpoint(latest)
                self._loaded_slots = {}
            else:
                self._loaded_slots = self._model.load_weights(latest)
                it = self._model.loaded_extras.get("SGD/iter")
                self.optimizer.iterations = int(it) if it is not None else 0
            self.restore_status = _RestoreStatus(before - set(self._model.variables))
            return
        if "export" in self.run_mode:
            raise AssertionError("No checkpoints found in {}, aborting.".format(self.model_dir))
        logging.warning("No existing checkpoints found in %s, running model in %s mode with random weights "
                        "initialization!", self.model_dir, self.run_mode)

    def _save(self, name):
        path = os.path.join(self.model_dir, name)
        if self._engine is not None:
            self._engine.save_checkpoint(path)
        else:
            self._model.save_weights(path)
        return path

    def assign_moving_averaged_weights(self):
        """executor.py:259-272: swap the moving averages in; returns the non-averaged weights for the swap back."""
        if not self.params.training.optimizer.use_moving_average:
            raise AssertionError("Cannot assign moving average weights since `use_moving_average` flag is set to False")
        if self._engine is not None:
            with torch.cuda.device(self._model.device):
                self._engine.store_to_model(use_ema=False)
                non_averaged = self._model.get_weights()
                logging.info("Loading moving average weights into model")
                self._engine.store_to_model(use_ema=True)
            return non_averaged
        non_averaged = self._model.get_weights()
        logging.info("Loading moving average weights into model")
        for (var, slot), arr in self._loaded_slots.items():
            if slot == "average" and var in self._model.variables:
                v = self._model.variables[var]
                v.copy_(torch.as_tensor(np.asarray(arr)).reshape(v.shape))
        self._model._refresh()
        return non_averaged

    def dump_config(self, config_path=None):
        if config_path is None:
            config_path = os.path.join(self.model_dir, "{}.json".format(self.name))
        with open(config_path, "w") as f:
            f.write(json.dumps(self.params, indent=4))
        logging.info("Dumping config to %s", config_path)

    # ---- weight decay (executor.py:296-327) -------------------------------------------------------------------
    def _get_weight_decay_variables(self):
        names = []
        for layer in self._model.layers:
            if not layer.trainable:
                continue
            for w in layer.weights:
                last = w.name.rsplit("/", 1)[-1]
                if "kernel" in last or "weight" in last:      # kernel / depthwise_kernel / pointwise_kernel
                    names.append(w.name)
                else:
                    assert "normalization" in w.name or "bias" in last, w.name
        return names

    def weight_decay(self):
        """alpha * sum l2_loss(kernel) over the weight-decay variables (host-side logging value; inside the training
        step the same number comes out of the optimizer kernels as `l2-regularization`)"""
        alpha = self.params.training.weight_decay_alpha
        v = self._model.variables
        return float(sum(alpha * 0.5 * float((v[n].double() ** 2).sum()) for n in self._weight_decay_vars))

    # ---- steps (executor.py:385-453) -----------------------------------------------------------------------------
    def _eval_step(self, data):
        detections = self._eval_model(data["image"].to(self._model.device), training=False)
        return {"image_id": data["image_id"], "detections": detections, "resize_scale": data["resize_scale"]}

    def _pad_eval_batch(self, data, per_replica):
        """One process per GPU reads its own shard of the validation files, so the ranks' LAST batches differ: a rank
        may hold a short batch, or none while another still has records.  Every rank therefore always contributes a
        batch of the fixed per-replica size to the gather — missing rows are zero images — plus a 0/1 validity row
        mask; the gathered rows are filtered by it.  (The reference never meets this: one process feeds all replicas
        from an unsharded validation set, executor.py:455-552.)"""
        H, W = int(self.params.input.input_shape[0]), int(self.params.input.input_shape[1])
        n = 0 if data is None else int(torch.as_tensor(data["image"]).shape[0])
        if n > per_replica:
            raise ValueError(f"validation batch of {n} images on a replica whose share is {per_replica}")
        image = torch.zeros((per_replica, H, W, 3), dtype=torch.float32)
        image_id = torch.full((per_replica,), -1, dtype=torch.int64)
        scale = torch.ones((per_replica, 2), dtype=torch.float32)
        if n:
            image[:n] = torch.as_tensor(data["image"]).to(torch.float32).cpu()
            image_id[:n] = torch.as_tensor(data["image_id"]).to(torch.int64).reshape(-1).cpu()
            scale[:n] = torch.as_tensor(data["resize_scale"]).to(torch.float32).reshape(n, 2).cpu()
        mask = torch.zeros((per_replica,), dtype=torch.int32)
        mask[:n] = 1
        return {"image": image, "image_id": image_id, "resize_scale": scale}, mask

    def distributed_eval_step(self, data, row_mask=None):
        res = self._eval_step(data)
        st = self.distribute_strategy
        if self.num_replicas > 1:       # strategy.gather(axis=0) over the replicas (executor.py:397-398)
            dev = self._model.device
            res = {"image_id": st.gather(torch.as_tensor(res["image_id"]).to(dev)).cpu(),
                   "detections": {k: st.gather(v) for k, v in res["detections"].items()},
                   "resize_scale": st.gather(torch.as_tensor(res["resize_scale"]).to(dev)).cpu()}
            if row_mask is not None:    # drop the padding rows of short / exhausted shards
                keep = st.gather(row_mask.to(dev)).bool()
                res = {"image_id": res["image_id"][keep.cpu()],
                       "detections": {k: v[keep] for k, v in res["detections"].items()},
                       "resize_scale": res["resize_scale"][keep.cpu()]}
        return res

    def _gathered_eval_results(self, total_steps):
        """Yields the (gathered) results of one evaluation step after the other, at most `total_steps` of them.
        Multi-replica loop control is collective: the ranks agree each round whether ANY of them still holds records, so
        a rank whose shard is exhausted keeps taking part in the gathers (with an all-padding batch) instead of leaving
        the others blocked in all_gather; `total_steps` comes from the config and is the same on every rank."""
        multi = self.num_replicas > 1
        per_replica = max(self.batch_size["val"] // self.num_replicas, 1)
        iterator, steps = iter(self._val_dataset()), 0
        while steps < total_steps:
            data = next(iterator, None)
            row_mask = None
            if multi:
                if not self.distribute_strategy.any_true(data is not None):
                    break
                data, row_mask = self._pad_eval_batch(data, per_replica)
            elif data is None:
                break
            steps += 1
            yield self.distributed_eval_step(data, row_mask)

    def _train_step(self, data):
        images, targets = data
        return self._engine.train_step(images.to(self._model.device), targets)

    def distributed_train_step(self, iterator, num_steps):
        loss = None
        for _ in range(int(num_steps)):
            loss = self._train_step(next(iterator))
        self._engine.finish_step()      # loss scale / optimizer.iterations of the last step (read right after this call)
        keys = sorted(k for k, v in loss.items() if hasattr(v, "item") or isinstance(v, (int, float)))
        packed = torch.stack([torch.as_tensor(_f(loss[k]) if not hasattr(loss[k], "reshape") else loss[k],
                                              dtype=torch.float32, device=self._model.device).reshape(())
                              for k in keys])
        packed = self.distribute_strategy.reduce_mean(packed)     # ReduceOp.MEAN over the replicas (:450-452)
        vals = packed.cpu().tolist()
        return dict(zip(keys, vals))

    # ---- evaluation (executor.py:455-552) --------------------------------------------------------------------------
    def continuous_evaluate(self, sleep_time=60, max_rounds=None):
        current, rounds = None, 0
        while max_rounds is None or rounds < max_rounds:
            latest = tf_checkpoint.latest_checkpoint(self.model_dir)
            if latest and latest != current:
                self._restore_checkpoint(latest)
                self.restore_status.assert_consumed()
                self.evaluate()
                current = latest
            rounds += 1
            if max_rounds is not None and rounds >= max_rounds:
                break
            logging.info("Sleeping for %s secs before checking for new checkpoint", sleep_time)
            sleep(sleep_time)

    def evaluate(self):
        if "eval" not in self._summary_writers:
            self._setup_summary_writers()
        non_averaged = None
        if self.params.training.optimizer.use_moving_average:
            non_averaged = self.assign_moving_averaged_weights()
        elif self._engine is not None:
            self._engine.store_to_model(use_ema=False)
        total_steps = self.val_steps
        current_step = int(self.optimizer.iterations)
        evaluator = COCOEvaluator(input_shape=self.params.input.input_shape,
                                  annotation_file_path=self.params.training.annotation_file_path,
                                  prediction_file_path=self.name + ".json",
                                  remap_class_ids=self.params.training.remap_class_ids)
        logging.info("Evaluating at step %d for %d steps", current_step, total_steps)
        meter = AverageMeter("eval_steps_per_second")
        multi = self.num_replicas > 1
        for i, results in enumerate(self._gathered_eval_results(total_steps)):
            start = time()
            if len(results["image_id"]) and self._is_chief:     # only the chief evaluates: the others need no copy
                evaluator.accumulate_results(results)
            execution_time = max(np.round(time() - start, 2), 1e-2)
            meter.accumulate(1 / execution_time)
            sps = meter.averaged_value
            logging.info("[global_step %d/%d][eval_step %d/%d] [ETA: %s] [%.2f imgs/s]", current_step, self.train_steps,
                         i + 1, total_steps, format_eta((total_steps - (i + 1)) / sps), sps * self.batch_size["val"])
        # every rank holds the same gathered detections; the chief alone writes the prediction file and runs COCOeval
        # (all ranks opening `<name>.json` with 'w' and reading it back is a truncate / read race), then shares the scores
        # A failure on the chief (no detections, COCOeval, file I/O) must not leave the other ranks waiting in the broadcast:
        # the chief sends (ok, scores | error text) and every rank raises together.
        scores, err = None, None
        if self._is_chief:
            try:
                scores = evaluator.evaluate()
            except Exception as e:   # noqa: BLE001 — re-raised on every rank below
                if not multi:
                    raise
                err = f"{type(e).__name__}: {e}"
        if multi:
            ok, payload = self.distribute_strategy.broadcast_object((err is None, scores if err is None else err), src=0)
            if not ok:
                raise RuntimeError(f"evaluation failed on the chief replica: {payload}")
            scores = payload
        if "eval" in self._summary_writers:
            self._summary_writers["eval"].scalars(current_step, {k: scores[k] for k in (
                "AP-IoU=0.50:0.95", "AP-IoU=0.50", "AP-IoU=0.75", "AR-(all)-IoU=0.50:0.95", "AR-(L)-IoU=0.50:0.95")})
        logging.info("[trial %d/%d][global_step %d/%d] evaluation results: %s", self._current_trial, self._max_trials,
                     current_step, self.train_steps, {k: float(np.round(v, 3)) for k, v in scores.items()})
        if non_averaged is not None:
            logging.info("Loading back non averaged weights into model")
            self._model.set_weights(non_averaged)
        return scores

    # ---- training loop (executor.py:571-744) -------------------------------------------------------------------------
    def _run_training_loop(self):
        if self.restore_checkpoint and self.restore_status is not None:
            self.restore_status.assert_consumed()
        start_step = int(self.optimizer.iterations)
        current_step = start_step
        if "val" in self.run_mode:
            logging.info("Running evaluation every %s steps", self.val_freq)
        if current_step >= self.train_steps:
            logging.info("Training completed at step %d", current_step)
            return True
        logging.info("Starting training from step %d for %d steps with %d steps per execution", start_step,
                     self.train_steps, self.steps_per_execution)
        if not self._save_during_training:
            logging.warning("Saving checkpoints only after completing training!")
        else:
            logging.info("Saving checkpoints every %d steps in %s", self.save_every, self.model_dir)
        if "train" not in self._summary_writers:
            self._setup_summary_writers()
        if self.use_float16:
            logging.info("Training with AMP turned on!")
        iterator = iter(self._train_dataset())
        meter = AverageMeter(name="train_steps_per_second")
        for _ in range(start_step, self.train_steps, self.steps_per_execution):
            start = time()
            n = min(self.steps_per_execution, self.train_steps - current_step)
            loss_dict = self.distributed_train_step(iterator, n)
            current_step = int(self.optimizer.iterations)
            torch.cuda.synchronize(self._model.device)
            end = time()
            loss_dict["execution-time"] = float(max(np.round(end - start, 2), 1e-2))
            loss_dict["learning-rate"] = float(self.optimizer.lr(current_step))
            meter.accumulate(n / loss_dict["execution-time"])
            sps = meter.averaged_value
            eta = format_eta((self.train_steps - current_step) / sps)
            if self._save_during_training and current_step % self.save_every == 0 and self._is_chief:
                logging.info("Saving checkpoint at step %d", current_step)
                self._save("weights_step_{}".format(current_step))
            if "train" in self._summary_writers:
                self._summary_writers["train"].scalars(current_step, loss_dict)
            logging.info("[trial: %d/%d][global_step %d/%d][ETA: %s][%.2f imgs/s] %s", self._current_trial,
                         self._max_trials, current_step, self.train_steps, eta, sps * self.batch_size["train"],
                         {k: float(np.round(v, 4)) for k, v in loss_dict.items()})
            if self.params.training.recovery.use_inflection_detector:
                if self._inflection_detector.is_value_anomalous(loss_dict[self.params.training.recovery.metric_key]):
                    logging.warning("Found inflection in %s values!, recent values: %s", self._inflection_detector.name,
                                    self._inflection_detector.data[-5:])
                    self._current_trial += 1
                    return False
            if self.val_freq > 0 and current_step % self.val_freq == 0 and not self._run_evaluation_at_end \
                    and "val" in self.run_mode:
                self.evaluate()
        if self._is_chief:
            logging.info("Saving final checkpoint at step %d", current_step)
            self._save("final_weights_step_{}".format(current_step))
        if self._run_evaluation_at_end and "val" in self.run_mode:
            self.evaluate()
        self._current_trial += 1
        return True

    def train(self):
        done = self._run_training_loop()
        while not done and self._current_trial < self._max_trials:
            self.distribute_strategy.barrier()     # the chief's last _save is complete before anyone looks
            latest = tf_checkpoint.latest_checkpoint(self.model_dir)
            if self.num_replicas > 1:               # one answer for every rank
                latest = self.distribute_strategy.broadcast_object(latest, src=0)
            if latest is not None:
                at = int(latest.split("_")[-1])
                resume_at = self.save_every * ((at // self.save_every) - 1)
                if resume_at == 0:
                    break
                self._restore_checkpoint(checkpoint=os.path.join(self.model_dir, "weights_step_{}".format(resume_at)))
            if self.params.training.recovery.use_inflection_detector:
                self._inflection_detector.reset()
            done = self._run_training_loop()
        if not done:
            logging.warning("Training failed after %d tries", self._current_trial)

    def get_flops(self):
        """multiply-accumulates of one inference forward pass (executor.py:754-770), counted from the static graph"""
        g, total = self._model.graph, 0
        for op in g.ops:
            if op["op"] in ("conv", "stem"):
                c = g.convs[op["conv"]]
                H, W, _, _ = g.tensors[op["out"]]
                total += H * W * c["k"] * c["k"] * c["cin"] * c["cout"]
            elif op["op"] == "dwconv":
                d = g.dws[op["dw"]]
                H, W, _, _ = g.tensors[op["out"]]
                total += H * W * d["k"] * d["k"] * d["C"]
        return int(total)

    @property
    def model(self):
        return self._model

    @property
    def weight_decay_variables(self):
        return self._weight_decay_vars
