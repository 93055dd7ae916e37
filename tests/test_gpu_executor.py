"""GPU: the reference's entry points end to end on a tiny problem — `Executor(...).run()` training a
few steps with checkpoints and an evaluation pass, resume, the `python -m retinanet` and `python -m retinanet.export`
command lines, the exported `serving_default` / `prepare_image` signatures, `model(images, training=True)` and
`normalize_image`."""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SIZE = 128


def _dataset(tmp_path):
    """16 PNG samples in 4 train + 4 val tfrecord shards and a COCO-style annotation file for the val images"""
    from test_tfrecord_cpu import _png_encode
    from retinanet.dataset_utils.tfrecord_writer import TFrecordWriter
    rng = np.random.default_rng(5)
    cats = [{"id": 10 + 3 * i, "name": "class-%02d" % (79 - i)} for i in range(80)]    # sorted-name remap is non-trivial
    images, anns = [], []
    for split in ("train", "val"):
        w = TFrecordWriter(16, 4, output_dir=str(tmp_path), prefix=split)
        for i in range(16):
            h, wd = int(rng.integers(60, 100)), int(rng.integers(60, 100))
            img = rng.integers(0, 256, size=(h, wd, 3)).astype(np.uint8)
            n = int(rng.integers(1, 4))
            lo = rng.uniform(0.05, 0.5, size=(n, 2))
            boxes = np.concatenate([lo, lo + rng.uniform(0.2, 0.45, size=(n, 2))], axis=1).astype(np.float32)
            classes = rng.integers(0, 80, size=(n,))
            image_id = 500 + i
            w.push(_png_encode(img, [i % 5]), boxes, classes, image_id)
            if split == "val":
                images.append({"id": image_id, "height": h, "width": wd})
                for b, c in zip(boxes, classes):
                    x1, y1, x2, y2 = b[0] * wd, b[1] * h, b[2] * wd, b[3] * h
                    anns.append({"id": len(anns) + 1, "image_id": image_id, "category_id": sorted(cats, key=lambda t: t["name"])[int(c)]["id"],
                                 "bbox": [float(x1), float(y1), float(x2 - x1), float(y2 - y1)], "area": float((x2 - x1) * (y2 - y1)),
                                 "iscrowd": 0})
        w.flush_last()
    ann_path = tmp_path / "instances_val.json"
    ann_path.write_text(json.dumps({"images": images, "annotations": anns, "categories": cats}))
    return str(ann_path)


def _params(tmp_path, ann_path):
    from retinanet.cfg import default_params
    p = default_params(input_size=SIZE, batch_train=4, batch_val=2, inference_batch=2)
    p.architecture.backbone.depth = 14
    p.architecture.batch_norm.use_sync = False
    p.experiment.name = "tiny"
    p.experiment.run_mode = "train_val"
    p.experiment.model_dir = str(tmp_path / "model_files")
    p.experiment.tensorboard_dir = str(tmp_path / "tensorboard")
    p.training.train_steps = 3
    p.training.steps_per_execution = 2
    p.training.save_every = 2
    p.training.validation_samples = 4
    p.training.validation_freq = -1
    p.training.restore_checkpoint = True
    p.training.freeze_variables = []
    p.training.annotation_file_path = ann_path
    p.training.recovery.use_inflection_detector = False
    p.inference.score_threshold = 0.001
    p.dataloader_params.tfrecords = {"train": str(tmp_path / "train-*"), "val": str(tmp_path / "val-*")}
    p.dataloader_params.shuffle_buffer_size = 8
    return p


@pytest.fixture(scope="module")
def trained(tmp_path_factory, cuda):
    from retinanet import Executor
    from retinanet.dataloader import InputPipeline
    from retinanet.distribute import get_strategy
    from retinanet.model import ModelBuilder
    tmp_path = tmp_path_factory.mktemp("exec")
    ann = _dataset(tmp_path)
    p = _params(tmp_path, ann)
    cwd = os.getcwd()
    os.chdir(tmp_path)          # the evaluator writes `<name>.json` next to the working directory, like the reference
    try:
        strategy = get_strategy(p.training.strategy)
        ex = Executor(params=p, strategy=strategy, run_mode="train_val", model_builder=ModelBuilder(p, "train_val", device=cuda),
                      train_input_fn=InputPipeline("train", p, False, 1, device=cuda),
                      val_input_fn=InputPipeline("val", p, False, 1, device=cuda))
        w0 = ex._engine.P.clone()
        ex.run()
    finally:
        os.chdir(cwd)
    return tmp_path, p, ex, w0


def test_executor_trains_checkpoints_and_evaluates(trained):
    from retinanet import tf_checkpoint
    tmp_path, p, ex, w0 = trained
    assert int(ex.optimizer.iterations) == 3 and ex._engine.step_count == 3
    assert not torch.equal(w0, ex._engine.P) and torch.isfinite(ex._engine.P).all()
    mdir = os.path.join(p.experiment.model_dir, p.experiment.name)
    assert os.path.exists(os.path.join(mdir, "weights_step_2.index"))          # save_every = 2
    assert os.path.exists(os.path.join(mdir, "final_weights_step_3.index"))
    assert tf_checkpoint.latest_checkpoint(mdir).endswith("final_weights_step_3")
    assert not os.path.exists(os.path.join(mdir, "tiny.json"))     # dump_config: run_mode == 'train' only (executor.py:267-268)
    rows = [json.loads(l) for l in open(os.path.join(p.experiment.tensorboard_dir, "tiny", "train", "scalars.jsonl"))]
    assert [r["step"] for r in rows] == [2, 3]                                  # steps_per_execution = 2, then the rest
    for key in ("box-loss", "class-loss", "weighted-loss", "total-loss", "l2-regularization", "gradient-norm",
                "num-anchors-matched", "learning-rate", "execution-time"):
        assert key in rows[0] and np.isfinite(rows[0][key]), key
    assert rows[0]["total-loss"] == pytest.approx(rows[0]["weighted-loss"] + rows[0]["l2-regularization"], rel=1e-5)
    assert rows[0]["l2-regularization"] == pytest.approx(ex.weight_decay(), rel=0.02)      # logged term vs host recompute
    ev = [json.loads(l) for l in open(os.path.join(p.experiment.tensorboard_dir, "tiny", "eval", "scalars.jsonl"))]
    assert ev and ev[-1]["step"] == 3 and 0.0 <= ev[-1]["AP-IoU=0.50:0.95"] <= 1.0
    preds = json.load(open(os.path.join(tmp_path, "tiny.json")))               # the prediction dump of the evaluator
    assert preds and {"image_id", "category_id", "bbox", "score"} <= set(preds[0])
    assert all(500 <= d["image_id"] < 516 and d["category_id"] in range(10, 250, 3) for d in preds)


def test_executor_resumes_where_it_stopped(trained, cuda):
    from retinanet import Executor
    from retinanet.dataloader import InputPipeline
    from retinanet.distribute import get_strategy
    from retinanet.model import ModelBuilder
    tmp_path, p, ex, _ = trained
    p2 = copy.deepcopy(p)
    p2.training.train_steps = 4
    ex2 = Executor(params=p2, strategy=get_strategy(p2.training.strategy), run_mode="train",
                   model_builder=ModelBuilder(p2, "train", device=cuda),
                   train_input_fn=InputPipeline("train", p2, False, 1, device=cuda))
    assert int(ex2.optimizer.iterations) == 3                                   # restored from final_weights_step_3
    torch.testing.assert_close(ex2._engine.P, ex._engine.P, rtol=0, atol=0)
    torch.testing.assert_close(ex2._engine.V, ex._engine.V, rtol=0, atol=0)
    mdir = os.path.join(p.experiment.model_dir, p.experiment.name)
    assert json.load(open(os.path.join(mdir, "tiny.json")))["training"]["train_steps"] == 4     # dump_config in 'train' mode
    ex2.run()
    assert int(ex2.optimizer.iterations) == 4


def test_command_lines_and_export_signatures(trained, cuda, tmp_path):
    from retinanet import export
    from retinanet.__main__ import main as cli_main
    from retinanet.dataloader.preprocessing_pipeline import PreprocessingPipeline
    from retinanet.model import ModelBuilder
    src, p, ex, _ = trained
    cfg = tmp_path / "config_tiny.json"      # (the evaluator dumps its predictions to <experiment.name>.json in the cwd)
    cfg.write_text(json.dumps(p))
    # python -m retinanet --config_path ... --run_evaluation : evaluation only, from the latest checkpoint
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        ex_eval = cli_main(["--config_path", str(cfg), "--run_evaluation", "--global_seed=3"])
        from retinanet import tf_checkpoint
        latest = tf_checkpoint.latest_checkpoint(os.path.join(p.experiment.model_dir, p.experiment.name))
        assert ex_eval.run_mode == "val" and int(ex_eval.optimizer.iterations) == int(latest.rsplit("_", 1)[1])
        # python -m retinanet.export --mode tf --export_saved_model --export_checkpoint
        out_dir = tmp_path / "export"
        export.main(["--config_path", str(cfg), "--mode", "tf", "--export_dir", str(out_dir), "--export_saved_model",
                     "--export_checkpoint"])
    finally:
        os.chdir(cwd)
    sm_dir = out_dir / "tiny" / "tf"
    assert (out_dir / "tiny" / "config.json").exists() and (sm_dir / "weights.safetensors").exists()
    assert (out_dir / "tiny" / (os.path.basename(latest) + ".index")).exists()
    spec = json.load(open(sm_dir / "signatures.json"))["signatures"]
    assert spec["serving_default"]["inputs"]["image"]["shape"] == [2, SIZE, SIZE, 3]
    assert spec["serving_default"]["outputs"]["boxes"]["shape"] == [2, 100, 4]
    sm = export.load(str(sm_dir), device=cuda)
    assert set(sm.signatures) == {"serving_default", "prepare_image"}
    raw = torch.rand((70, 90, 3), generator=torch.Generator().manual_seed(1)) * 255
    prepared = sm.signatures["prepare_image"](image=raw)["image"]
    assert tuple(prepared.shape) == (1, SIZE, SIZE, 3)
    want = PreprocessingPipeline(p.input.input_shape, p.dataloader_params).normalize_and_resize_with_pad(raw)["image"]
    torch.testing.assert_close(prepared[0], want, rtol=0, atol=0)
    batch = torch.cat([prepared, prepared.flip(2)]).contiguous()
    det = sm.signatures["serving_default"](image=batch)
    assert det["boxes"].shape == (2, 100, 4) and det["classes"].dtype == torch.int32 and det["valid_detections"].shape == (2,)
    # the export holds the MOVING AVERAGES (use_moving_average: true) — same detections as the EMA weights served directly
    b = ModelBuilder(p, "val", device=cuda)
    m = b()
    slots = m.load_weights(latest)
    for (var, slot), arr in slots.items():
        if slot == "average":
            m.variables[var].copy_(torch.as_tensor(np.asarray(arr)).reshape(m.variables[var].shape))
    m._refresh()
    ref = b.add_post_processing_stage(m)(batch.to(cuda))
    for k in ("boxes", "scores", "classes", "valid_detections"):
        torch.testing.assert_close(det[k], ref[k], rtol=0, atol=0)
    with pytest.raises(ValueError):
        sm.signatures["serving_default"](image=batch[:1])          # the batch size is part of the signature


def test_model_call_training_flag_and_normalize_image(cuda):
    from retinanet.cfg import default_params
    from retinanet.dataloader import normalize_image
    from retinanet.model import ModelBuilder
    p = default_params(input_size=SIZE)
    p.architecture.backbone.depth = 14
    p.architecture.batch_norm.use_sync = False
    model = ModelBuilder(p, "train", device=cuda)()
    images = torch.randn((2, SIZE, SIZE, 3), generator=torch.Generator().manual_seed(2)).to(cuda)
    tr = {k: {lv: t.clone() for lv, t in d.items()} for k, d in model(images, training=True).items()}
    inf = model(images, training=False)
    assert set(tr) == {"class-predictions", "box-predictions"} and sorted(tr["box-predictions"]) == list("34567")
    assert tr["class-predictions"]["3"].shape == (2, SIZE // 8, SIZE // 8, 720) and tr["class-predictions"]["3"].dtype == torch.float32
    # batch statistics vs moving statistics (0 / 1 at initialisation): the two modes differ
    assert (tr["box-predictions"]["3"] - inf["box-predictions"]["3"]).abs().max().item() > 1e-3
    eng = model.train_engine(2)
    again = eng.forward(images)
    torch.testing.assert_close(again["box-predictions"]["4"], tr["box-predictions"]["4"], rtol=0, atol=0)
    # normalize_image (dataloader/utils.py:58-66): (x / pixel_scale - mean) / stddev, float32 op by op
    pp = p.dataloader_params.preprocessing
    img = (torch.rand((37, 53, 3), generator=torch.Generator().manual_seed(3)) * 255).float()
    got = normalize_image(img, pp.mean, pp.stddev, pp.pixel_scale).cpu().numpy()
    x = img.numpy()
    want = ((x / np.float32(pp.pixel_scale)) - np.asarray(pp.mean, np.float32)) / np.asarray(pp.stddev, np.float32)
    np.testing.assert_array_equal(got, want.astype(np.float32))
