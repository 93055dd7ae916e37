"""librnet_hip_f16.so — the library built from the same sources with -DRN_F16: IEEE-half storage and
v_mfma_f32_32x32x16_f16, the arithmetic of the reference's `mixed_float16` policy (BASELINE config 5: EfficientNet-B3,
fp16 mixed precision + LossScaleOptimizer).

Kernel-level parity of the half build = the cases of tests/test_gpu_conv.py and tests/test_gpu_train_kernels.py that
are parametrized with build="f16" (half tensors against the same float64 restatement, whose rounding points then round
to half): forward kernels on all three kernel families, split weight planes, weight / data gradients, BatchNorm passes
and their fusions, pooling / FPN backward, the optimizer.  Network level: tests/test_gpu_efficientnet.py (its configs
carry `mixed_float16`), the `mixed_float16` cases of tests/test_gpu_train_step.py, and the ResNet case below."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_builds_with_their_storage_types(cuda):
    from retinanet import _C
    assert _C.lib().rn_storage_dtype() == 0 and _C.lib(True).rn_storage_dtype() == 1
    for name in _C.exported_symbols():
        assert hasattr(_C.lib(True), name), name


def test_resnet_forward_under_mixed_float16(cuda):
    """ResNet-50 RetinaNet under the mixed_float16 policy: the engine picks the half build, the restatement rounds to
    half at the same layer boundaries; same bounds as the bfloat16 model test (half has three more mantissa bits)."""
    from model_ref import RefModel
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    import test_gpu_model as tm
    p = default_params(input_size=128, precision="mixed_float16")
    model = ModelBuilder(p, "val", device=cuda, seed=5)()
    eng = model.inference_engine(2)
    assert eng.f16 and eng.lib.rn_storage_dtype() == 1
    g = torch.Generator().manual_seed(1)
    images = torch.randn((2, 128, 128, 3), generator=g)
    preds = eng(images.to(cuda))
    ref = RefModel(p, model.variables, emulate_bf16=True)(images)
    tm._check_predictions(preds, ref)
