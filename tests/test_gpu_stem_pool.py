"""rn_stem_conv_bn_relu_pool (the ResNet stem in one launch) against the launches it replaces: rn_conv2d_nhwc_fwd on the
packed NHWC4 input (R=7, S=1, Cin=32, stride 2; folded BatchNorm + relu) followed by rn_maxpool2d_nhwc 3x3/2 SAME —
bit for bit (same K order, same rounding points), and against a float64 evaluation of resnet.py:288-307."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("N,H,W,act", [(2, 64, 64, "relu"), (1, 100, 72, "relu"), (2, 250, 250, "relu"),
                                       (1, 130, 258, "relu6"), (2, 640, 640, "relu")])
def test_fused_stem_equals_conv_then_pool(cuda, N, H, W, act):
    from retinanet import _C
    lib = _C.lib()
    st = _C.current_stream()
    g = torch.Generator().manual_seed(H * 1000 + W)
    img = torch.randn((N, H, W, 3), generator=g).to(cuda)
    w = (torch.randn((7, 7, 3, 64), generator=g) * 0.1).to(cuda).contiguous()
    scale = (torch.rand((64,), generator=g) + 0.5).to(cuda)
    scale[5] = -0.7           # a negative BatchNorm scale
    shift = (torch.randn((64,), generator=g) * 0.3).to(cuda)
    Hs, Ws = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Po, Qo = -(-Hs // 2), -(-Ws // 2)
    pt = max((Po - 1) * 2 + 3 - Hs, 0) // 2
    pl = max((Qo - 1) * 2 + 3 - Ws, 0) // 2
    Hp = max((Hs - 1) * 2 + 7, H + 3)
    Wp = -(-max((Ws - 1) * 2 + 8, W + 3) // 8) * 8
    xin = torch.empty((N, Hp, Wp, 4), dtype=torch.bfloat16, device=cuda)
    wp = torch.empty((64, 7, 32), dtype=torch.bfloat16, device=cuda)
    _C.check(lib.rn_pack_stem_weight_rs(_C.ptr(w), 7, 7, 64, _C.ptr(wp), st))
    _C.check(lib.rn_pack_image_nhwc4(_C.ptr(img), N, H, W, 3, 3, Hp, Wp, _C.ptr(xin), st))
    # the two launches
    y = torch.empty((N, Hs, Ws, 64), dtype=torch.bfloat16, device=cuda)
    p = _C.ConvProblem()
    p.R, p.S, p.stride_h, p.stride_w, p.pad_top, p.pad_left = 7, 1, 2, 2, 0, 0
    p.act, p.out_dtype, p.num_segments = _C.ACT_IDS[act], _C.RN_DT_BF16, 1
    s = p.seg[0]
    s.x, s.w, s.y, s.scale, s.shift, s.residual = xin.data_ptr(), wp.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None
    s.N, s.H, s.W, s.Cin, s.pix_stride, s.Ho, s.Wo, s.Cout = N, Hp, Wp, 32, 4, Hs, Ws, 64
    _C.check(lib.rn_conv2d_nhwc_fwd(ctypes.byref(p), st))
    want = torch.empty((N, Po, Qo, 64), dtype=torch.bfloat16, device=cuda)
    _C.check(lib.rn_maxpool2d_nhwc(_C.ptr(y), _C.ptr(want), N, Hs, Ws, 64, 3, 2, pt, pl, Po, Qo, st))
    # one launch
    got = torch.full((N, Po, Qo, 64), -7.0, dtype=torch.bfloat16, device=cuda)
    _C.check(lib.rn_stem_conv_bn_relu_pool(_C.ptr(xin), _C.ptr(wp), _C.ptr(scale), _C.ptr(shift), _C.ptr(got), N, Hp, Wp,
                                           Hs, Ws, 7, 64, _C.ACT_IDS[act], 3, 2, pt, pl, Po, Qo, st))
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert float(want.float().abs().max()) > 1.0
    if H <= 130:   # float64 evaluation of the layer sequence on the bf16 operands
        xi = F.pad(_bf(img).double().permute(0, 3, 1, 2), (3, 3, 3, 3))
        z = F.conv2d(xi, _bf(w).double().permute(3, 2, 0, 1), stride=2)
        z = _bf(z.float()).double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        z = torch.clamp(z, min=0.0) if act == "relu" else torch.clamp(z, 0.0, 6.0)
        z = F.pad(z, (pl, 2 * (Qo - 1) + 3 - Ws - pl, pt, 2 * (Po - 1) + 3 - Hs - pt), value=float("-inf"))
        ref = F.max_pool2d(z, 3, 2).permute(0, 2, 3, 1)
        torch.testing.assert_close(got.double(), ref, rtol=2 ** -7, atol=2e-2)


def test_other_shapes_are_refused(cuda):
    from retinanet import _C
    lib = _C.lib()
    x = torch.zeros((1, 70, 72, 4), dtype=torch.bfloat16, device=cuda)
    w = torch.zeros((64, 7, 32), dtype=torch.bfloat16, device=cuda)
    y = torch.zeros((1, 16, 16, 64), dtype=torch.bfloat16, device=cuda)
    st = _C.current_stream()
    # 3x3 stem (EfficientNet), 32 output channels, swish: not this kernel
    assert lib.rn_stem_conv_bn_relu_pool(_C.ptr(x), _C.ptr(w), None, None, _C.ptr(y), 1, 70, 72, 32, 32, 3, 64, _C.ACT_IDS["relu"],
                                         3, 2, 0, 0, 16, 16, st) != 0
    assert lib.rn_stem_conv_bn_relu_pool(_C.ptr(x), _C.ptr(w), None, None, _C.ptr(y), 1, 70, 72, 32, 32, 7, 32, _C.ACT_IDS["relu"],
                                         3, 2, 0, 0, 16, 16, st) != 0
    assert lib.rn_stem_conv_bn_relu_pool(_C.ptr(x), _C.ptr(w), None, None, _C.ptr(y), 1, 70, 72, 32, 32, 7, 64, _C.ACT_IDS["swish"],
                                         3, 2, 0, 0, 16, 16, st) != 0
    assert "ResNet stem" in lib.rn_last_error().decode()
