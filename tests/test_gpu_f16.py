"""librnet_hip_f16.so — the library built from the same sources with -DRN_F16: IEEE-half storage and
v_mfma_f32_32x32x16_f16, the arithmetic of the reference's `mixed_float16` policy (BASELINE config 5: EfficientNet-B3,
fp16 mixed precision + LossScaleOptimizer).  The kernel tests of tests/test_gpu_conv.py are re-run on it with half
tensors against the same float64 restatement (whose rounding points then round to half); the network-level parity of
the half build is tests/test_gpu_efficientnet.py, whose configs carry `mixed_float16`, and the ResNet case below."""
import pytest
import torch

import test_gpu_conv as tc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def half(monkeypatch):
    monkeypatch.setattr(tc, "H16", torch.float16)
    yield


def test_two_builds_with_their_storage_types(cuda):
    from retinanet import _C
    assert _C.lib().rn_storage_dtype() == 0 and _C.lib(True).rn_storage_dtype() == 1
    for name in _C.exported_symbols():
        assert hasattr(_C.lib(True), name), name


@pytest.mark.parametrize("case", tc.CASES, ids=lambda c: "f16-" + "x".join(str(v) for v in c))
def test_conv_128_row_kernel_f16(cuda, half, case):
    tc.test_conv_single(cuda, case)


@pytest.mark.parametrize("case", [c for c in tc.CASES if c[4] > 128 and c[4] % 8 == 0][:6],
                         ids=lambda c: "f16-big-" + "x".join(str(v) for v in c))
def test_conv_256_row_kernels_f16(cuda, half, case):
    from retinanet import _C
    lib = _C.lib(True)
    lib.rn_debug_conv_tile(2)
    try:
        tc.test_conv_single(cuda, case)
    finally:
        lib.rn_debug_conv_tile(0)


@pytest.mark.parametrize("case", tc.HALO_CASES[:6], ids=lambda c: "f16-halo-" + "x".join(str(v) for v in c))
def test_conv_halo_kernel_f16(cuda, half, case):
    tc.test_conv_halo_kernel(cuda, case)


def test_split_weight_planes_f16(cuda, half):
    """the float32 prediction convs: two / three half planes of the f32 kernel"""
    for case in tc.SPLIT_CASES[:4]:
        tc.test_conv_f32_weights_as_split_bf16_planes(cuda, case)


# ---- the training kernels (tests/test_gpu_train_kernels.py) on the half build ----------------------------------------
import test_gpu_train_kernels as tk


@pytest.fixture()
def half_train(monkeypatch):
    monkeypatch.setattr(tk, "H16", torch.float16)
    yield


def test_wgrad_kernels_f16(cuda, half_train, request):
    for shape in tk.WGRAD_SHAPES[:5] + tk.WGRAD_SHAPES[7:10]:   # 128-tile kernel and wgrad_big_kernel
        tk.test_wgrad(cuda, shape, request)


@pytest.mark.parametrize("k,stride,cin,cout", [(3, 1, 128, 256), (1, 1, 256, 128), (3, 2, 128, 128), (1, 2, 256, 512)])
def test_dgrad_f16(cuda, half_train, k, stride, cin, cout):
    tk.test_dgrad_via_forward_kernel(cuda, k, stride, cin, cout)


def test_dgrad_subpixel_f16(cuda, half_train):
    tk.test_dgrad_stride2_subpixel(cuda, 1, 20, 16, 128, 128, True)


@pytest.mark.parametrize("act,use_res", [("relu", True), ("relu", False), ("swish", False)])
def test_batchnorm_train_kernels_f16(cuda, half_train, act, use_res):
    tk.test_bn_train_forward_backward(cuda, act, use_res)


@pytest.mark.parametrize("k,tile", [(1, 2), (3, 2), (3, 1)])
def test_batchnorm_fusions_f16(cuda, half_train, k, tile):
    tk.test_bn_forward_stats_fused_into_conv_epilogue(cuda, k, tile)
    tk.test_bn_backward_reduction_fused_into_the_data_gradient(cuda, k, tile)
    tk.test_bn_backward_gate_from_the_bit_mask(cuda, "relu")


def test_pool_topdown_balance_backward_f16(cuda, half_train):
    tk.test_pool_topdown_balance_backward(cuda)


def test_optimizer_step_f16(cuda, half_train):
    tk.test_optimizer_step(cuda)


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
