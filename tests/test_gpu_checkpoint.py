"""GPU: SURVEY §8(f)-3 on the live objects — `RetinaNetModel.save_weights/load_weights` in TensorFlow's checkpoint
format and `TrainEngine.save_checkpoint/restore_checkpoint` (executor.py:221-244, 652-654): an interrupted run
resumed from its checkpoint must land exactly where the uninterrupted run does (every kernel is deterministic)."""
import numpy as np
import pytest
import torch

from test_gpu_train_step import _setup

pytestmark = pytest.mark.gpu


def test_model_weights_roundtrip_and_same_predictions(cuda, tmp_path):
    from retinanet import tf_checkpoint as ck
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    p = default_params(input_size=128)
    a = ModelBuilder(p, "train", device=cuda, seed=5)()
    b = ModelBuilder(p, "train", device=cuda, seed=6)()
    k0 = next(k for k in a.variables if k.endswith("/kernel"))
    assert not torch.equal(a.variables[k0], b.variables[k0])
    prefix = str(tmp_path / "model_files" / "weights_step_128")
    a.save_weights(prefix)
    assert ck.latest_checkpoint(tmp_path / "model_files") == prefix
    r = ck.TensorBundleReader(prefix)
    assert r.entries[k0 + "/.ATTRIBUTES/VARIABLE_VALUE"]["shape"] == tuple(a.variables[k0].shape)   # HWIO, as Keras
    b.load_weights(prefix)
    for k in a.variables:
        assert torch.equal(a.variables[k], b.variables[k]), k
    x = torch.randn((1, 128, 128, 3), generator=torch.Generator().manual_seed(0)).to(cuda)
    pa, pb = a(x, training=False), b(x, training=False)
    for lvl in pa["class-predictions"]:
        assert torch.equal(pa["class-predictions"][lvl], pb["class-predictions"][lvl])
    with pytest.raises(KeyError):          # a checkpoint without some variable
        ck.save_weights(str(tmp_path / "partial"), {k0: a.variables[k0].cpu().numpy()})
        b.load_weights(str(tmp_path / "partial"))
    b.load_weights(str(tmp_path / "partial"), skip_mismatch=True)
    a.save_weights(str(tmp_path / "w.safetensors"))
    b.load_weights(str(tmp_path / "w.safetensors"))


def test_resume_from_checkpoint_is_bit_exact(cuda, tmp_path):
    from retinanet.optimizers import build_optimizer

    def fresh(seed):
        p, model, eng, targets, images = _setup(cuda, 128, 2, True, seed=seed, depth=26)
        p.training.optimizer.lr_params.warmup_learning_rate = 0.01
        p.training.optimizer.lr_params.initial_learning_rate = 0.02
        model.optimizer = build_optimizer(p.training.optimizer, p.training.train_steps, p.floatx.precision)
        return model, eng, targets, images.to(cuda)

    model, eng, targets, images = fresh(7)
    for _ in range(2):
        eng.train_step(images, targets)
    prefix = str(tmp_path / "weights_step_2")
    eng.save_checkpoint(prefix)
    want = [eng.train_step(images, targets)["weighted-loss"].item() for _ in range(2)]
    torch.cuda.synchronize()
    want_P, want_V, want_E = eng.P.clone(), eng.V.clone(), eng.E.clone()
    want_mm = {bn: d["mm"].clone() for bn, d in eng.bn_state.items()}

    # a new process would build the model from another seed and restore
    model2, eng2, targets2, images2 = fresh(7)
    with torch.no_grad():
        for v in model2.variables.values():
            v.add_(0.01)
    eng2.load_from_model()
    eng2.restore_checkpoint(prefix)
    assert eng2.step_count == 2 and model2.optimizer.iterations == 2
    got = [eng2.train_step(images2, targets2)["weighted-loss"].item() for _ in range(2)]
    torch.cuda.synchronize()
    assert got == want
    assert torch.equal(eng2.P, want_P) and torch.equal(eng2.V, want_V) and torch.equal(eng2.E, want_E)
    for bn, mm in want_mm.items():
        assert torch.equal(eng2.bn_state[bn]["mm"], mm)
