"""`python -m retinanet --config_path <json> [--model_dir ...] ...` — the reference's training / evaluation entry point
(retinanet/__main__.py:15-171) with the same flag names and the same order of overrides, driving the MI355X engines.

Launch: one process per GPU.  `training.strategy.type: "gpu"` runs on one MI355X; `"multi_gpu"` expects to be started
by `python -m torch.distributed.run --nproc-per-node N -m retinanet ...` (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) and runs data parallel over RCCL.  `--xla` and `--gpu_memory_allow_growth` are accepted for command-line
compatibility and do nothing here (there is no tracing compiler and the engines allocate static buffers)."""
from __future__ import annotations

import logging
import os
import sys

from retinanet.flags import FlagSet

SUPPORTED_RUN_MODES = ["train", "val", "train_val", "continuous_eval"]


def define_flags():
    f = FlagSet()
    f.DEFINE_integer("global_seed", 1337, "Sets global seed for all random ops")
    f.DEFINE_string("config_path", None, "Path to the config file")
    f.DEFINE_string("model_dir", None, "Overides `model_dir` specified in the config")
    f.DEFINE_string("resume_from", None, "Overides latest_checkpoint")
    f.DEFINE_boolean("enable_weights_info", False, "Write histogram and norm for each trainable weight")
    f.DEFINE_boolean("run_evaluation", False, "Overides `run_mode` specified in the config")
    f.DEFINE_boolean("run_continuous_evaluation", False, "Overides `run_mode` specified in the config")
    f.DEFINE_boolean("xla", False, "Compile with XLA JIT (accepted, unused)")
    f.DEFINE_boolean("gpu_memory_allow_growth", False, "(accepted, unused)")
    f.DEFINE_boolean("is_multi_host", False, "Set this to true if running a multi-node setup")
    f.DEFINE_boolean("debug", False, "Print debugging info")
    f.DEFINE_string("log_dir", None, "absl's log directory flag")
    return f


def main(argv=None):
    FLAGS = define_flags().parse(argv)
    import torch
    from retinanet import Executor
    from retinanet.cfg import Config
    from retinanet.dataloader import InputPipeline
    from retinanet.distribute import get_strategy
    from retinanet.model import ModelBuilder

    torch.manual_seed(FLAGS.global_seed)
    logging.basicConfig(level=logging.DEBUG if FLAGS.debug else logging.INFO,
                        format="%(levelname).1s %(asctime)s %(filename)s:%(lineno)d] %(message)s")
    if FLAGS.config_path is None:
        raise SystemExit("--config_path is required")
    params = Config(FLAGS.config_path).params
    if FLAGS.log_dir and not os.path.exists(FLAGS.log_dir):
        os.makedirs(FLAGS.log_dir, exist_ok=True)
    if FLAGS.log_dir:
        logging.getLogger().addHandler(logging.FileHandler(os.path.join(FLAGS.log_dir, params.experiment.name + ".log")))
    logging.warning("Using %d as global seed", FLAGS.global_seed)
    if FLAGS.is_multi_host:
        logging.warning("Running in multi_host mode")
    if FLAGS.xla:
        logging.warning("--xla: there is no tracing compiler in this build; ignored")
    logging.info("Compute dtype: %s", {"mixed_bfloat16": "bfloat16", "mixed_float16": "float16"}.get(
        str(params.floatx.precision), "float32"))
    logging.info("Variable dtype: float32")

    strategy = get_strategy(params.training.strategy)
    logging.info("Running on %d replicas", strategy.num_replicas_in_sync)
    run_mode = params.experiment.run_mode
    if FLAGS.run_evaluation:
        logging.warning("Overiding `run_mode` from %s to evaluation only", run_mode)
        run_mode = "val"
    if FLAGS.run_continuous_evaluation:
        logging.warning("Overiding `run_mode` from %s to continuous evaluation", run_mode)
        run_mode = "continuous_eval"
    if run_mode not in SUPPORTED_RUN_MODES:
        raise AssertionError("Unsupported run mode requested, available run modes: {}".format(SUPPORTED_RUN_MODES))
    if FLAGS.model_dir is not None:
        logging.warning("Overiding `model_dir` from %s to %s", params.experiment.model_dir, FLAGS.model_dir)
        params.experiment.model_dir = FLAGS.model_dir

    train_input_fn = val_input_fn = None
    if "train" in run_mode:
        train_input_fn = InputPipeline(run_mode="train", params=params, is_multi_host=FLAGS.is_multi_host,
                                       num_replicas=strategy.num_replicas_in_sync, device=strategy.device)
    if "val" in run_mode or run_mode == "continuous_eval":
        val_input_fn = InputPipeline(run_mode="val", params=params, is_multi_host=FLAGS.is_multi_host,
                                     num_replicas=strategy.num_replicas_in_sync, device=strategy.device)
    model_builder = ModelBuilder(params, run_mode=run_mode, device=strategy.device, seed=FLAGS.global_seed)
    executor = Executor(params=params, strategy=strategy, run_mode=run_mode, model_builder=model_builder,
                        train_input_fn=train_input_fn, val_input_fn=val_input_fn, is_multi_host=FLAGS.is_multi_host,
                        enable_weights_info=FLAGS.enable_weights_info, resume_from=FLAGS.resume_from)
    executor.run()
    return executor


if __name__ == "__main__":
    main(sys.argv[1:])
