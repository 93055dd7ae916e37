"""`python -m retinanet.export --config_path <json> --mode tf --export_saved_model` — the reference's export entry point
(retinanet/export.py:19-351) and the loader of what it writes.

The reference freezes the Keras model into a TensorFlow SavedModel with two signatures (export.py:194-208, 244-270):
  serving_default(image: f32[inference.batch_size, H, W, 3]) -> {boxes f32[B,100,4] (normalised x1,y1,x2,y2),
      scores f32[B,100], classes i32[B,100], valid_detections i32[B]}
  prepare_image(image: f32[h, w, 3]) -> {image: f32[1, H, W, 3]}   (normalise, aspect-preserving bilinear resize, zero pad)
A SavedModel protobuf cannot be written without TensorFlow, and the graph it would hold is TensorFlow ops — the thing
this build replaces.  What is exported instead, under `<export_dir>/<experiment.name>/`:
  config.json                      the config (as export.py:163 dumps it)
  <mode>/weights.safetensors       every variable under its Keras name, conv kernels HWIO float32 (moving averages
                                   swapped in unless --ignore_moving_average_weights)
  <mode>/signatures.json           the two signatures' input / output specs
and `retinanet.export.load(<export_dir>/<name>/<mode>)` returns an object whose `.signatures['serving_default']` /
`.signatures['prepare_image']` are callables with the reference's keyword argument (`image=`) and output dicts, running
the HIP engines (the forward pass + post-processing of `serving_default` replays as ONE captured HIP graph).
Modes: `tf` is served; `tf_tensorrt`, `onnx`, `onnx_tensorrt` need TF-TRT / tf2onnx / TensorRT (NVIDIA tooling): out of
scope, they raise.  `--export_h5` needs h5py, which this image does not have: it raises; `--export_checkpoint` writes the
TensorFlow checkpoint format (retinanet.tf_checkpoint)."""
from __future__ import annotations

import json
import logging
import os
import shutil
import sys

from retinanet.flags import FlagSet


def define_flags():
    f = FlagSet()
    f.DEFINE_string("config_path", None, "Path to the config file", required=True)
    f.DEFINE_enum("mode", None, ["tf", "tf_tensorrt", "onnx", "onnx_tensorrt"],
                  "Export mode for `saved_models`. Controls skipping decoding/NMS stages", required=True)
    f.DEFINE_string("export_dir", "export", "Path to store the model artefacts")
    f.DEFINE_boolean("export_saved_model", False, "Export weights as a `saved_model`")
    f.DEFINE_boolean("export_h5", False, "Export weights as an h5 file (can be used for fine tuning)")
    f.DEFINE_boolean("export_checkpoint", False, "Export weights in tensorflow object checkpoint format")
    f.DEFINE_string("checkpoint_name", "latest", "Restores model weights from `checkpoint_name`")
    f.DEFINE_boolean("ignore_moving_average_weights", False,
                     "Loads non averaged weights if `use_moving_average` is set to True")
    f.DEFINE_string("model_dir", None, "Overides `model_dir` specified in the config")
    f.DEFINE_boolean("skip_prepare_image_fn", False, "Skip exporting `prepare_image` signature")
    f.DEFINE_boolean("debug", False, "Print debugging info")
    f.DEFINE_string("log_dir", None, "absl's log directory flag")
    # TensorRT calibration flags of the reference: accepted so that existing command lines parse
    f.DEFINE_string("calibration_images_dir", "coco/val2017", "Calibration images dir")
    f.DEFINE_enum("calibration_method", "entropy", ["entropy", "minmax"], "INT8 Calibration method")
    f.DEFINE_integer("calibration_batch_size", 8, "Batch size for calibration")
    f.DEFINE_integer("num_calibration_images", 5000, "Number of images used in calibration")
    f.DEFINE_string("precision", "fp32", "Execution precision for TensorRT Engines")
    return f


def signature_specs(params, with_prepare_image=True):
    H, W = params.input.input_shape
    B, D = int(params.inference.batch_size), int(params.inference.max_detections)
    specs = {"serving_default": {
        "inputs": {"image": {"dtype": "float32", "shape": [B, H, W, int(params.input.channels)]}},
        "outputs": {"boxes": {"dtype": "float32", "shape": [B, D, 4]}, "scores": {"dtype": "float32", "shape": [B, D]},
                    "classes": {"dtype": "int32", "shape": [B, D]}, "valid_detections": {"dtype": "int32", "shape": [B]}}}}
    if with_prepare_image:
        specs["prepare_image"] = {"inputs": {"image": {"dtype": "float32", "shape": [None, None, 3]}},
                                  "outputs": {"image": {"dtype": "float32", "shape": [1, H, W, 3]}}}
    return specs


def write_saved_model(model, params, out_dir, mode="tf", with_prepare_image=True):
    from safetensors.torch import save_file
    if mode != "tf":
        raise NotImplementedError(f"export mode {mode!r} needs TF-TRT / tf2onnx / TensorRT: out of scope on MI355X")
    if os.path.exists(out_dir):
        logging.warning("Found existing artefacts in %s, clearing old files", out_dir)
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    save_file({k: v.detach().cpu().contiguous() for k, v in model.variables.items()},
              os.path.join(out_dir, "weights.safetensors"))
    with open(os.path.join(out_dir, "signatures.json"), "w") as f:
        json.dump({"format": "retinanet-mi355x-saved-model", "version": 1, "mode": mode,
                   "signatures": signature_specs(params, with_prepare_image)}, f, indent=2)
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        f.write(json.dumps(params, indent=4))
    return out_dir


class SavedModel:
    """What `load()` returns: `.signatures[name](image=...)` like `tf.saved_model.load(path).signatures[name]`."""

    def __init__(self, path, device=None):
        import torch
        from retinanet.cfg import AttrDict
        from retinanet.dataloader.preprocessing_pipeline import PreprocessingPipeline
        from retinanet.model import ModelBuilder
        with open(os.path.join(path, "signatures.json")) as f:
            self.meta = json.load(f)
        if self.meta.get("format") != "retinanet-mi355x-saved-model":
            raise ValueError(f"{path} is not an export of this build")
        with open(os.path.join(path, "config.json")) as f:
            self.params = AttrDict(json.load(f))
        builder = ModelBuilder(self.params, run_mode="export", device=device)
        self.model = builder()
        self.model.load_weights(os.path.join(path, "weights.safetensors"))
        self._infer = builder.prepare_model_for_export(self.model, mode=self.meta["mode"])
        self._pre = PreprocessingPipeline(self.params.input.input_shape, self.params.dataloader_params)
        spec = self.meta["signatures"]
        shape = tuple(spec["serving_default"]["inputs"]["image"]["shape"])

        def serving_default(image):
            image = torch.as_tensor(image, dtype=torch.float32).to(self.model.device)
            if tuple(image.shape) != shape:
                raise ValueError(f"serving_default expects image of shape {shape}, got {tuple(image.shape)}")
            return self._infer(image.contiguous())

        def prepare_image(image):
            out = self._pre.normalize_and_resize_with_pad(torch.as_tensor(image, dtype=torch.float32))
            return {"image": out["image"].unsqueeze(0)}

        self.signatures = {"serving_default": serving_default}
        if "prepare_image" in spec:
            self.signatures["prepare_image"] = prepare_image


def load(path, device=None):
    return SavedModel(path, device=device)


def main(argv=None):
    FLAGS = define_flags().parse(argv)
    from retinanet import Executor
    from retinanet.cfg import Config
    from retinanet.distribute import Strategy
    from retinanet.model import ModelBuilder
    import torch

    logging.basicConfig(level=logging.DEBUG if FLAGS.debug else logging.INFO)
    params = Config(FLAGS.config_path).params
    if FLAGS.model_dir:
        params.experiment.model_dir = FLAGS.model_dir
        logging.warning("Using local path %s as `model_dir`", params.experiment.model_dir)
    checkpoint_name = None if FLAGS.checkpoint_name == "latest" else FLAGS.checkpoint_name
    run_mode = "export"
    params.architecture.backbone.checkpoint = ""       # skip loading pretrained backbone weights
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    model_builder = ModelBuilder(params, run_mode=run_mode, device=device)
    executor = Executor(params=params, strategy=Strategy("gpu", device), run_mode=run_mode, model_builder=model_builder,
                        train_input_fn=None, val_input_fn=None, resume_from=checkpoint_name)
    export_dir = os.path.join(FLAGS.export_dir, params.experiment.name)
    saved_model_export_dir = os.path.join(export_dir, FLAGS.mode)
    os.makedirs(export_dir, exist_ok=True)
    executor.dump_config(os.path.join(export_dir, "config.json"))
    executor.restore_status.assert_consumed()
    if params.training.optimizer.use_moving_average:
        non_averaged = executor.assign_moving_averaged_weights()
        if FLAGS.ignore_moving_average_weights:
            logging.warning("Loading back non averaged weights into model")
            executor.model.set_weights(non_averaged)
    from retinanet import tf_checkpoint
    if FLAGS.export_h5:
        raise NotImplementedError("--export_h5 needs h5py, which is not available here; use --export_checkpoint "
                                  "(TensorFlow checkpoint format) or --export_saved_model (safetensors)")
    if FLAGS.export_checkpoint:
        latest = os.path.basename(tf_checkpoint.latest_checkpoint(executor.model_dir))
        export_file_path = os.path.join(export_dir, latest)
        logging.info("Exporting weights in tensorflow checkpoint format to %s", export_file_path)
        executor.model.save_weights(export_file_path)
    if FLAGS.export_saved_model:
        logging.info("Exporting `saved_model` to %s", FLAGS.export_dir)
        with_pre = not FLAGS.skip_prepare_image_fn and "tf" in FLAGS.mode
        if not with_pre:
            logging.warning("Skipping `prepare_image` signature in `saved_model`")
        write_saved_model(executor.model, params, saved_model_export_dir, mode=FLAGS.mode, with_prepare_image=with_pre)
        for name, spec in signature_specs(params, with_pre).items():
            logging.info("\nSignature: %s\n Input Shapes:\n %s\nOutput Shapes:\n%s", name,
                         {k: v["shape"] for k, v in spec["inputs"].items()},
                         {k: v["shape"] for k, v in spec["outputs"].items()})
    return executor


if __name__ == "__main__":
    main(sys.argv[1:])
