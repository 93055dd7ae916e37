"""GPU: SURVEY §8(f)-4 end to end — TFRecord shards written by `TFrecordWriter` flow through `InputPipeline`
(records -> parse_example -> preprocessing on the GPU -> batch -> `encode_batch`) and come out as what the
oracle's restatement of the reference's per-sample map function gives
(dataloader/preprocessing_pipeline.py:13-129, label_encoder.py:89-125, input_pipeline.py:27-92)."""
import copy

import numpy as np
import pytest
import torch

import oracle as o
from test_tfrecord_cpu import _png_encode

pytestmark = pytest.mark.gpu


class _FixedDraws:
    """Stands in for the pipeline's numpy Generator: hands out the given draws in order."""

    def __init__(self, draws):
        self.draws = list(draws)

    def uniform(self, lo=0.0, hi=1.0, size=None):
        v = self.draws.pop(0)
        return np.asarray(v, dtype=np.float64) if size is not None else float(v)


def _params(params, tmp_path, size, aug, batch):
    p = copy.deepcopy(params)
    p.input.input_shape = [size, size]
    p.training.batch_size = {"train": batch, "val": batch}
    p.dataloader_params["tfrecords"] = {"train": str(tmp_path / "train-*"), "val": str(tmp_path / "val-*")}
    p.dataloader_params["shuffle_buffer_size"] = 1
    p.dataloader_params["augmentations"] = aug
    return p


def _samples(n, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        h, w = int(rng.integers(40, 120)), int(rng.integers(40, 120))
        img = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
        k = int(rng.integers(0, 5))
        lo = rng.uniform(0, 0.6, size=(k, 2))
        boxes = np.concatenate([lo, np.minimum(lo + rng.uniform(0.05, 0.5, size=(k, 2)), 1.0)], axis=1).astype(np.float32)
        out.append((img, boxes, rng.integers(0, 80, size=(k,)).astype(np.int64), 500 + i))
    return out


def _write(tmp_path, prefix, samples, shards):
    from retinanet.dataset_utils.tfrecord_writer import TFrecordWriter
    w = TFrecordWriter(len(samples), shards, output_dir=str(tmp_path), prefix=prefix)
    for (img, boxes, classes, image_id) in samples:
        w.push(_png_encode(img, [1, 4, 2]), boxes, classes, image_id)
    w.flush_last()


@pytest.mark.parametrize("draws", [
    dict(flip=0.9, scale=1.7, off=(0.3, 0.8)),     # flipped, scaled past the canvas: crop at an offset
    dict(flip=0.2, scale=0.45, off=(0.5, 0.5)),    # not flipped, shrunk: pad
    dict(flip=0.7, scale=1.0, off=(0.0, 0.99)),
])
def test_train_sample_bit_exact_with_given_draws(cuda, params, tmp_path, draws):
    from retinanet.dataloader.preprocessing_pipeline import PreprocessingPipeline
    aug = {"use_augmentation": True, "horizontal_flip": True, "scale_jitter": {"min_scale": 0.1, "max_scale": 2.0}}
    p = _params(params, tmp_path, 128, aug, 2)
    pipe = PreprocessingPipeline(p.input.input_shape, p.dataloader_params)
    pp = p.dataloader_params.preprocessing
    for (img, boxes, classes, _) in _samples(4, seed=11):
        pipe.rng = _FixedDraws([draws["flip"], draws["scale"], draws["off"]])
        image, b, c = pipe({"image": img.astype(np.float32), "objects": {"bbox": boxes, "label": classes}})
        torch.cuda.synchronize()
        want_img, want_b, want_c = o.train_preprocess(img, boxes, classes, [128, 128], pp.mean, pp.stddev,
                                                      pp.pixel_scale, draws["flip"] > 0.5, draws["scale"], draws["off"])
        np.testing.assert_array_equal(image.cpu().numpy().view(np.uint32), want_img.view(np.uint32))
        np.testing.assert_array_equal(b, want_b)
        np.testing.assert_array_equal(c, want_c)
        assert image.shape == (128, 128, 3) and b.dtype == np.float32 and c.dtype == np.int32


def test_train_batches_match_oracle_targets(cuda, params, tmp_path):
    from retinanet.dataloader.input_pipeline import InputPipeline
    samples = _samples(7)
    _write(tmp_path, "train", samples, 1)
    p = _params(params, tmp_path, 128, {"use_augmentation": False}, 3)
    pipe = InputPipeline("train", p, False, 1, device=cuda)
    pipe.cycle_length = 1
    it = pipe()
    pp = p.dataloader_params.preprocessing
    anchors = o.generate_anchors(128, 128, 3, 7, p.anchor_params.areas, p.anchor_params.aspect_ratios, p.anchor_params.scales)
    # one file, shuffle buffer 1, no augmentation: samples come in file order; 7 samples -> 2 full batches per epoch
    # (drop_remainder), the third batch starts with sample 6 and wraps into the next epoch (repeat)
    order = [0, 1, 2, 3, 4, 5, 6, 0, 1]
    for bi in range(3):
        images, targets = next(it)
        torch.cuda.synchronize()
        assert images.shape == (3, 128, 128, 3) and images.is_cuda
        for j in range(3):
            img, boxes, classes, _ = samples[order[bi * 3 + j]]
            want_img, want_b, want_c = o.train_preprocess(img, boxes, classes, [128, 128], pp.mean, pp.stddev,
                                                          pp.pixel_scale, False, None, None)
            np.testing.assert_array_equal(images[j].cpu().numpy().view(np.uint32), want_img.view(np.uint32))
            matches, cls_t, box_t, _ = o.encode_sample(anchors, want_b, want_c.astype(np.float32))
            np.testing.assert_array_equal(targets["_flat"]["class-targets"][j].cpu().numpy(), cls_t)
            np.testing.assert_array_equal(targets["_flat"]["box-targets"][j].cpu().numpy().view(np.uint32),
                                          np.asarray(box_t, np.float32).view(np.uint32))
            assert float(targets["num-positives"][j]) == float((np.asarray(matches) > -1).sum())
        assert targets["class-targets"]["3"].shape == (3, 16, 16, 9) and targets["box-targets"]["7"].shape == (3, 1, 1, 36)


def test_val_batches(cuda, params, tmp_path):
    from retinanet.dataloader.input_pipeline import InputPipeline
    samples = _samples(5, seed=8)
    _write(tmp_path, "val", samples, 2)
    p = _params(params, tmp_path, 96, {"use_augmentation": False}, 2)
    pipe = InputPipeline("val", p, False, 1, device=cuda)
    pp = p.dataloader_params.preprocessing
    by_id = {s[3]: s for s in samples}
    batches = list(pipe())
    assert [int(b["image"].shape[0]) for b in batches] == [2, 2, 1]          # drop_remainder=False
    seen = []
    for b in batches:
        for j, image_id in enumerate(b["image_id"].tolist()):
            img = by_id[image_id][0]
            want, scale = o.prepare_image(img.astype(np.float32), 96, 96, pp.mean, pp.stddev, pp.pixel_scale)
            np.testing.assert_array_equal(b["image"][j].cpu().numpy().view(np.uint32), want.view(np.uint32))
            np.testing.assert_array_equal(b["resize_scale"][j].numpy(), scale)
            seen.append(image_id)
    assert sorted(seen) == sorted(by_id)


def test_training_steps_from_tfrecords(cuda, params, tmp_path):
    """The whole chain on real files: TFRecord shards -> InputPipeline (parse, augment, batch, GPU label encoding)
    -> TrainEngine.train_step.  Losses must be finite and fall over a few steps on the repeated tiny dataset."""
    from retinanet.dataloader.input_pipeline import InputPipeline
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    from retinanet.optimizers import build_optimizer
    samples = _samples(6, seed=21)
    _write(tmp_path, "train", samples, 2)
    aug = {"use_augmentation": True, "horizontal_flip": True, "scale_jitter": {"min_scale": 0.8, "max_scale": 1.25}}
    p = _params(params, tmp_path, 128, aug, 2)
    p.architecture.batch_norm.use_sync = False
    p.architecture.backbone.depth = 26
    p.training.optimizer.lr_params.warmup_learning_rate = 0.01
    p.training.optimizer.lr_params.initial_learning_rate = 0.02
    builder = ModelBuilder(p, "train", device=cuda, seed=3)
    model = builder()
    model.optimizer = build_optimizer(p.training.optimizer, p.training.train_steps, p.floatx.precision)
    eng = TrainEngine(model, 2, frozen_regexes=[])
    pipe = InputPipeline("train", p, False, 1, device=cuda)
    it = pipe()
    losses = []
    for _ in range(12):
        images, targets = next(it)
        losses.append(eng.train_step(images, targets)["weighted-loss"].item())
    assert all(np.isfinite(losses)), losses
    assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses


def test_eval_chain_from_tfrecords(cuda, params, tmp_path):
    """executor._eval_step + COCOEvaluator on real files: val TFRecords -> InputPipeline('val') -> serving model ->
    COCOEvaluator.accumulate_results: one detection list per image id, boxes in ORIGINAL-image pixels (the
    resize_scale of the pipeline undone, int32 truncation), bit-exact against the oracle's accumulation."""
    from retinanet.dataloader.input_pipeline import InputPipeline
    from retinanet.eval import COCOEvaluator
    from retinanet.model import ModelBuilder
    samples = _samples(5, seed=31)
    _write(tmp_path, "val", samples, 1)
    p = _params(params, tmp_path, 128, {"use_augmentation": False}, 2)
    p.inference.score_threshold = 0.005
    b = ModelBuilder(p, "val", device=cuda, seed=4)
    model = b()
    infer = {bs: b.add_post_processing_stage(model) for bs in (1, 2)}
    ev = COCOEvaluator([128, 128], categories=[{"id": i + 1, "name": f"c{i:02d}"} for i in range(80)],
                       remap_class_ids=True)
    hw = {s[3]: s[0].shape[:2] for s in samples}
    seen = []
    for batch in InputPipeline("val", p, False, 1, device=cuda)():
        det = infer[int(batch["image"].shape[0])](batch["image"])
        res = {"image_id": batch["image_id"].numpy(), "resize_scale": batch["resize_scale"].numpy(), "detections": det}
        n0 = len(ev.processed_detections)
        ev.accumulate_results(res)
        valid = det["valid_detections"].cpu().numpy()
        wb, wc = o.coco_accumulate(det["boxes"].cpu().numpy(), det["classes"].cpu().numpy(), valid,
                                   batch["resize_scale"].numpy(), (128, 128), class_lut=ev._lut)
        k = n0
        for i, image_id in enumerate(batch["image_id"].tolist()):
            seen.append(image_id)
            for d in range(int(valid[i])):
                r = ev.processed_detections[k]
                k += 1
                assert r["image_id"] == image_id and r["bbox"] == wb[i, d].tolist() and r["category_id"] == int(wc[i, d])
                h, w = hw[image_id]
                # original-image pixels: inside the padded canvas (128 canvas pixels / resize_scale), which covers
                # the image; a randomly initialised net may put boxes on the padding
                sc = batch["resize_scale"].numpy()[i]
                assert -2 <= r["bbox"][0] <= 128 / sc[1] + 2 and -2 <= r["bbox"][1] <= 128 / sc[0] + 2
        assert k == len(ev.processed_detections)
    assert sorted(seen) == sorted(hw)
