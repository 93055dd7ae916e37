"""GPU parity: K1/K2 implicit-GEMM conv (+ bias / folded BN / residual / activation, with a bf16 tensor where the
reference has one between two layers: include/rnet_hip.h rn_conv_segment) and the pooling / FPN / BalanceFeatures
kernels against a float64 PyTorch-CPU restatement on the same bf16-rounded inputs.  Because the rounding points
are shared, the only difference left is fp32 summation order: an output may sit ONE bf16 step off where the sum
straddles a rounding boundary (rtol 2^-7), and only a small fraction of the outputs may."""
import ctypes
import math
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


# The 16-bit storage type under test.  A test parametrized with build="f16" runs on librnet_hip_f16.so (the same sources
# built with -DRN_F16: IEEE-half storage, v_mfma_f32_32x32x16_f16 — the arithmetic of `mixed_float16`) with half tensors
# against the same float64 restatement, whose rounding points then round to half; every other test runs the bfloat16
# build.  The autouse fixture below sets it per test from the test's own parameters: no test calls another test.
H16 = torch.bfloat16
_DT = {"bf16": torch.bfloat16, "f16": torch.float16}


@pytest.fixture(autouse=True)
def _storage_type(request):
    global H16
    params = request.node.callspec.params if hasattr(request.node, "callspec") else {}
    H16 = _DT[params.get("build", "bf16")]
    yield
    H16 = torch.bfloat16


def _builds(cases, f16_cases, tag=""):
    """(build, case) parameters: every case on the bfloat16 build + `f16_cases` on the half build"""
    ps = [pytest.param("bf16", c, id=tag + "x".join(str(v) for v in c)) for c in cases]
    return ps + [pytest.param("f16", c, id="f16-" + tag + "x".join(str(v) for v in c)) for c in f16_cases]


def _lib():
    from retinanet import _C
    return _C.lib(H16 == torch.float16)


def _bf(x):
    return x.to(H16)


def _seed(case):
    """reproducible across processes (hash() of a tuple with strings is salted per process)"""
    return zlib.crc32(repr(case).encode()) % (2 ** 31)


def _conv_gpu(cuda, segs, k, stride, pad, act, out_f32, opts=None, splitk_ws=None, expect_split=None):
    """opts: rn_launch_opts fields of THIS launch (kernel family, persistent grid, ...): per-call, no process state;
    splitk_ws: zero-filled uint8 device tensor handed to the launch as rn_conv_problem.splitk_ws"""
    from retinanet import _C
    lib = _lib()
    p = _C.ConvProblem()
    if opts:
        p.opts = _C.LaunchOpts(**opts)
    p.R = p.S = k
    p.stride_h = p.stride_w = stride
    p.pad_top = p.pad_left = pad
    p.act = _C.ACT_IDS[act]
    p.out_dtype = _C.RN_DT_F32 if out_f32 else _C.RN_DT_BF16
    p.num_segments = len(segs)
    keep, outs = [], []
    for i, s in enumerate(segs):
        x = _bf(s["x"]).to(cuda).contiguous()
        w = s["w"].to(cuda).float().contiguous()          # HWIO
        kk, _, cin, cout = w.shape
        cinp = lib.rn_conv_cin_pad(cin)
        terms = int(s.get("w_terms", 1))
        wp = torch.empty((lib.rn_conv_cout_pad(cout), kk, kk, terms * cinp), dtype=H16, device=cuda)
        if s.get("w_pair"):     # the two planes along Cout (rn_conv_segment.w_pair)
            assert terms == 1
            wp = torch.full((lib.rn_conv_pair_rows(cout), kk, kk, cinp), float("nan"), dtype=H16, device=cuda)
            _C.check(lib.rn_pack_conv_weight_pair(_C.ptr(w), 0, kk, kk, cin, cout, cinp, _C.ptr(wp), _C.current_stream()))
        elif terms > 1:
            _C.check(lib.rn_pack_conv_weight_split(_C.ptr(w), 0, kk, kk, cin, cout, cinp, terms, _C.ptr(wp),
                                                   _C.current_stream()))
        else:
            _C.check(lib.rn_pack_conv_weight(_C.ptr(w), kk, kk, cin, cout, cinp, _C.ptr(wp), _C.current_stream()))
        N, H, W, _ = x.shape
        Ho, Wo = (H + 2 * pad - kk) // stride + 1, (W + 2 * pad - kk) // stride + 1
        y = torch.empty((N, Ho, Wo, cout), dtype=torch.float32 if out_f32 else H16, device=cuda)
        sc = s.get("scale"); sh = s.get("shift"); res = s.get("residual"); bs = s.get("bias")
        sc = None if sc is None else sc.to(cuda).float().contiguous()
        sh = None if sh is None else sh.to(cuda).float().contiguous()
        bs = None if bs is None else bs.to(cuda).float().contiguous()
        res = None if res is None else _bf(res).to(cuda).contiguous()
        g = p.seg[i]
        g.x, g.w, g.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
        g.scale = sc.data_ptr() if sc is not None else None
        g.shift = sh.data_ptr() if sh is not None else None
        g.residual = res.data_ptr() if res is not None else None
        g.bias = bs.data_ptr() if bs is not None else None
        g.w_terms = terms
        g.w_pair = 1 if s.get("w_pair") else 0
        g.N, g.H, g.W, g.Cin, g.pix_stride, g.Ho, g.Wo, g.Cout = N, H, W, cin, cin, Ho, Wo, cout
        keep += [x, wp, sc, sh, res, bs]
        outs.append(y)
    if splitk_ws is not None:
        p.splitk_ws, p.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel()
    if expect_split is not None:
        assert (lib.rn_conv_splitk_workspace_bytes(ctypes.byref(p)) > 0) == expect_split
    _C.check(lib.rn_conv2d_nhwc_fwd(ctypes.byref(p), _C.current_stream()), "conv")
    torch.cuda.synchronize()
    return [y.float().cpu() for y in outs]


def _conv_ref(s, k, stride, pad, act, out_f32):
    """include/rnet_hip.h rn_conv_segment: bf16 output = Conv2D(+bias) -> bf16 -> BatchNorm affine -> bf16 ->
    residual add -> bf16 -> activation; f32 output (a dtype=float32 layer): f32 weights, no rounding."""
    rb = (lambda t: t) if out_f32 else (lambda t: _bf(t.float()).double())
    x = _bf(s["x"]).double().permute(0, 3, 1, 2)
    w = s["w"].double() if s.get("w_terms", 1) == 3 else _bf(s["w"]).double()
    if s.get("w_terms", 1) == 2:
        w = w + _bf(s["w"] - _bf(s["w"]).float()).double()
    y = F.conv2d(x, w.permute(3, 2, 0, 1), stride=stride, padding=pad).permute(0, 2, 3, 1)
    if s.get("bias") is not None:
        y = y + s["bias"].double()
    affine = s.get("scale") is not None or s.get("shift") is not None
    res = s.get("residual")
    if affine or res is not None:
        y = rb(y)
    if s.get("scale") is not None:
        y = y * s["scale"].double()
    if s.get("shift") is not None:
        y = y + s["shift"].double()
    if res is not None:
        y = (rb(y) if affine else y) + _bf(res).double()
    if act == "relu":
        y = F.relu(y)
    elif act == "relu6":
        y = F.relu6(y)
    elif act == "swish":
        y = rb(y)
        y = y * torch.sigmoid(y)
    return y.float() if out_f32 else _bf(y.float()).float()


def _close(got, want, out_f32):
    scale = want.abs().max().item() + 1e-6
    if out_f32:
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-4 * scale)
    else:
        torch.testing.assert_close(got, want, rtol=1.0 / 128, atol=scale / 256)
        assert (got - want).abs().mean().item() <= 2e-3 * scale
        # shared rounding points: all that is left is fp32 summation order flipping a rounding here and there
        # (an intermediate bf16 step shows up amplified by the BatchNorm scale in the few outputs it hits)
        assert (got != want).float().mean().item() < 0.03


CASES = [
    # N, H, W, Cin, Cout, k, stride, act, residual, out_f32
    (2, 20, 20, 64, 64, 1, 1, "relu", False, False),
    (2, 20, 20, 64, 256, 1, 1, "relu", True, False),
    (1, 40, 40, 256, 512, 1, 2, None, False, False),     # strided projection shortcut
    (2, 24, 24, 64, 64, 3, 1, "relu", False, False),
    (1, 40, 40, 128, 128, 3, 2, "relu", False, False),   # stride in the 3x3 (ResNet v1.5)
    (1, 10, 10, 512, 2048, 1, 1, "relu", True, False),
    (2, 10, 10, 256, 256, 3, 1, "relu6", False, False),
    (2, 5, 5, 256, 36, 3, 1, None, False, True),         # box prediction conv, f32 out, Cout pad 64
    (1, 10, 10, 256, 720, 3, 1, None, False, True),      # class prediction conv, 6 n-tiles
    (1, 13, 9, 192, 128, 3, 1, "relu", False, False),    # odd sizes, M tail, Cin = 3*64
    (1, 7, 7, 96, 64, 3, 1, None, False, False),         # Cin = 1.5 K steps: zero-padded weight tail
    (2, 9, 9, 40, 144, 1, 1, "swish", False, False),     # EfficientNet widths: Cin 40 (pad 64), Cout 144
    (1, 6, 6, 816, 136, 1, 1, None, True, False),        # MBConv project conv 816 -> 136 + skip
    (1, 8, 8, 24, 32, 3, 1, "relu", False, False),       # Cin 24 -> BK 32
    (2, 24, 24, 256, 256, 3, 1, "relu", True, False),    # 4.5 M tiles of 256, residual
    (3, 17, 19, 128, 512, 1, 1, "swish", False, False),  # 2 n-tiles of 256, M tail
    (1, 20, 20, 256, 720, 3, 1, None, False, True),      # class prediction conv, f32 out, 3 n-tiles of 256
]


def _run_conv_single(cuda, case, opts=None):
    N, H, W, Cin, Cout, k, stride, act, use_res, out_f32 = case
    g = torch.Generator().manual_seed(_seed(case))
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((k, k, Cin, Cout), generator=g) / math.sqrt(k * k * Cin),
         "scale": torch.rand((Cout,), generator=g) + 0.5, "shift": torch.randn((Cout,), generator=g) * 0.1}
    if use_res:
        s["residual"] = torch.randn((N, Ho, Wo, Cout), generator=g)
    got = _conv_gpu(cuda, [s], k, stride, pad, act, out_f32, opts)[0]
    _close(got, _conv_ref(s, k, stride, pad, act, out_f32), out_f32)


@pytest.mark.parametrize("build,case", _builds(CASES, CASES))
def test_conv_single(cuda, build, case):
    """what the dispatcher picks at these sizes: the 128-row conv_fwd_kernel"""
    _run_conv_single(cuda, case)


_WIDE_CASES = [c for c in CASES if c[4] > 64]


@pytest.mark.parametrize("build,case", _builds(_WIDE_CASES, _WIDE_CASES[:4], "wide-"))
def test_conv_single_full_width_tiles(cuda, build, case):
    """At these sizes (at most half a 128 x 128 tile per CU) the dispatcher narrows the 128-row kernel's tiles to 64
    columns; rn_launch_opts.conv_tile = 1 keeps the 128 x 128 form that larger launches run — the K order of an
    accumulator is the same, so the two agree bit for bit."""
    N, H, W, Cin, Cout, k, stride, act, use_res, out_f32 = case
    g = torch.Generator().manual_seed(_seed(case))
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((k, k, Cin, Cout), generator=g) / math.sqrt(k * k * Cin),
         "scale": torch.rand((Cout,), generator=g) + 0.5, "shift": torch.randn((Cout,), generator=g) * 0.1}
    if use_res:
        s["residual"] = torch.randn((N, Ho, Wo, Cout), generator=g)
    wide = _conv_gpu(cuda, [s], k, stride, pad, act, out_f32, dict(conv_tile=1))[0]
    auto = _conv_gpu(cuda, [s], k, stride, pad, act, out_f32, None)[0]
    _close(wide, _conv_ref(s, k, stride, pad, act, out_f32), out_f32)
    assert torch.equal(wide, auto)


_BIG_CASES = [c for c in CASES if c[4] > 128 and c[4] % 8 == 0]


@pytest.mark.parametrize("build,case", _builds(_BIG_CASES, _BIG_CASES[:6], "big-"))
def test_conv_big_tile_kernel(cuda, build, case):
    """Same cases through the 256-row persistent kernels (rn_conv_big.hip / rn_conv_halo.hip; normally picked only
    when the launch has >= 192 such tiles): rn_launch_opts.conv_tile = 2 on this launch."""
    _run_conv_single(cuda, case, opts=dict(conv_tile=2))


@pytest.mark.parametrize("build", ["bf16", "f16"])
def test_conv_grouped_pyramid_on_the_256_row_kernels(cuda, build):
    _run_grouped_pyramid(cuda, opts=dict(conv_tile=2))


HALO_CASES = [
    # N, H, W, Cin, Cout, act, residual, out_f32   (3x3 / stride 1 / pad 1, through conv_halo_kernel)
    (2, 24, 24, 256, 256, "relu", True, False),     # tiles 2 and 4 straddle the two images
    (1, 80, 80, 64, 256, "relu", False, False),     # widest level: 4 rows + halo per tile, 2 channel chunks
    (3, 5, 5, 256, 256, None, False, False),        # one tile over three tiny images (many pad rows)
    (7, 10, 10, 128, 320, "relu6", False, False),   # Cout tail: second n-tile has 64 live channels
    (1, 20, 20, 256, 720, None, False, True),       # class prediction conv, f32 out
    (2, 13, 9, 96, 512, "swish", True, False),      # odd sizes, Cin = 3 chunks, M tail
    (40, 3, 3, 32, 256, None, False, False),        # single chunk; 28 images per tile
    (1, 40, 40, 512, 512, "relu", False, False),    # ResNet stage-3 shape, 16 chunks
    (2, 1, 7, 64, 256, None, False, False),         # H = 1: every tap row but the middle one is padding
    (2, 6, 1, 64, 256, None, False, False),         # W = 1
]


@pytest.mark.parametrize("build,case", _builds(HALO_CASES, HALO_CASES[:6], "halo-"))
def test_conv_halo_kernel(cuda, build, case):
    """3x3 / stride 1 / pad 1 with Cout >= 256 through the halo-patch kernel (rn_conv_halo.hip), against the
    float64 reference and against conv_big_kernel on the same inputs (same products, other summation order)."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, act, use_res, out_f32 = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((3, 3, Cin, Cout), generator=g) / math.sqrt(9 * Cin),
         "scale": torch.rand((Cout,), generator=g) + 0.5, "shift": torch.randn((Cout,), generator=g) * 0.1}
    if use_res:
        s["residual"] = torch.randn((N, H, W, Cout), generator=g)
    p = _C.ConvProblem()
    p.R = p.S = 3
    p.stride_h = p.stride_w = p.pad_top = p.pad_left = 1
    p.out_dtype, p.num_segments = (_C.RN_DT_F32 if out_f32 else _C.RN_DT_BF16), 1
    sg = p.seg[0]
    sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout = N, H, W, Cin, Cin, H, W, Cout
    p.opts = _C.LaunchOpts(conv_tile=2)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == 2
    got = _conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, dict(conv_tile=2))[0]
    p.opts = _C.LaunchOpts(conv_tile=2, conv_no_halo=1)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == 1
    big = _conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, dict(conv_tile=2, conv_no_halo=1))[0]
    want = _conv_ref(s, 3, 1, 1, act, out_f32)
    _close(got, want, out_f32)
    scale = want.abs().max().item() + 1e-6
    # bf16 outputs may differ by one rounding step where the fp32 sums straddle a rounding boundary
    torch.testing.assert_close(got, big, rtol=1.0 / 128 if not out_f32 else 1e-4, atol=scale * (1 / 256 if not out_f32 else 1e-5))
    if not out_f32:
        assert (got != big).float().mean().item() < 0.02


SPLITK_CASES = [
    # N, H, W, Cin, Cout, act, residual, f32 output, bias, w_terms, persistent workgroups (0 = one per CU) -> tiles / rounds
    # (a launch splits when its tiles have >= 8 chunks of 32 input channels: >= 4 chunks per part)
    (2, 80, 80, 256, 256, "relu", False, False, False, 1, 0),    # 50 tiles on 256 CUs: one round, every tile in 2 parts
    (1, 40, 40, 256, 512, None, True, False, False, 1, 6),       # 14 tiles on 6 workgroups: 2 rounds + 2 tiles in 2 parts
    (1, 20, 20, 128, 720, None, False, True, False, 2, 5),       # f32 output, two weight planes: 6 tiles on 5, 1 tile in 2 parts
    (2, 24, 24, 256, 256, "relu", False, False, True, 1, 4),     # bias in the accumulators: part 0 only
    (3, 16, 16, 512, 256, "relu6", False, False, False, 1, 2),   # 3 tiles on 2: 1 round + 1 tile in 2 parts of 8 chunks
    (1, 30, 30, 512, 256, "relu", False, False, False, 1, 0),    # 4 tiles on 256 CUs: 4 parts of 4 chunks each
]


@pytest.mark.parametrize("build,case", _builds(SPLITK_CASES, SPLITK_CASES[:3], "splitk-"))
def test_conv_halo_split_last_round(cuda, build, case):
    """rn_conv_problem.splitk_ws: the tiles of a persistent launch's last round are cut along K over several workgroups
    (every part writes its fp32 accumulators to the workspace; per wave, the part that arrives last adds them in part order).  Same float64 reference as the whole-tile
    launch; the exchange adds in part order, so repeats are bit-identical; the workspace header is zero again afterwards
    (every arrival counter consumed by the last arriver)."""
    N, H, W, Cin, Cout, act, use_res, out_f32, use_bias, terms, wgs = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((3, 3, Cin, Cout), generator=g) / math.sqrt(9 * Cin), "w_terms": terms}
    if use_bias:
        s["bias"] = torch.randn((Cout,), generator=g)
    else:
        s["scale"], s["shift"] = torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.1
    if use_res:
        s["residual"] = torch.randn((N, H, W, Cout), generator=g)
    opts = dict(conv_tile=2, max_workgroups=wgs)
    ws = torch.zeros((80 << 20,), dtype=torch.uint8, device=cuda)
    want = _conv_ref(s, 3, 1, 1, act, out_f32)
    runs = [_conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, opts, splitk_ws=ws, expect_split=True)[0] for _ in range(3)]
    _close(runs[0], want, out_f32)
    for r in runs[1:]:
        assert torch.equal(r, runs[0])
    assert int(ws[:16384].view(torch.int32).abs().sum().item()) == 0
    whole = _conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, opts)[0]          # no workspace: whole tiles, another sum order
    _close(whole, want, out_f32)
    if not out_f32:
        assert (runs[0] != whole).float().mean().item() < 0.02


SPLIT128_CASES = [
    # N, H, W, Cin, Cout, k, stride, act, residual, out_f32, bias, w_terms, splitk_target_blocks   (the 128-row kernel, every tile cut along K)
    (1, 20, 20, 512, 512, 3, 1, "relu", False, False, False, 1, 0),     # ResNet stage 4 3x3 at batch 1: 32 tiles of 128 x 64, 72 K steps, 8 parts
    (1, 20, 20, 2048, 512, 1, 1, "relu", False, False, False, 1, 0),    # stage 4 first 1x1: 32 K steps, 8 parts of 4
    (1, 20, 20, 512, 2048, 1, 1, "relu", True, False, False, 1, 256),   # stage 4 last 1x1 + residual: 128 tiles, 8 K steps, 2 parts (at the default target of 192 workgroups it runs whole)
    (1, 40, 40, 256, 256, 3, 2, "relu6", False, False, False, 1, 0),    # stride 2; M tail (400 = 3 tiles + 16 pixels)
    (2, 13, 11, 192, 72, 3, 1, "swish", True, False, False, 1, 64),     # odd sizes, Cout tail, two images, swish + residual
    (1, 10, 10, 256, 36, 3, 1, None, False, True, True, 2, 0),          # f32 output, two weight planes, bias (box prediction, P6)
    (1, 25, 25, 96, 128, 1, 1, None, False, False, True, 1, 40),        # 96 channels (padded to 128: two K steps), bias only: too shallow, whole tiles
    (3, 9, 9, 1024, 256, 1, 1, "relu", False, False, False, 1, 24),     # 3 parts of 5 / 5 / 6 K steps (uneven shares)
]


@pytest.mark.parametrize("build,case", _builds(SPLIT128_CASES, SPLIT128_CASES[:4], "split128-"))
def test_conv_128_row_kernel_split_k(cuda, build, case):
    """rn_conv_problem.splitk_ws on the 128-row kernel (conv_fwd_kernel<..., SPLIT>): a small launch of a deep layer cuts
    every tile along K, each part writes its fp32 partial tile to the workspace and the part that arrives last adds them
    in part order and runs the epilogue.  Against the float64 reference; repeats are bit-identical (the sum order does not
    depend on who arrives last); the arrival counters are zero again afterwards; and the unsplit launch of the same
    problem (no workspace) differs only by summation order."""
    N, H, W, Cin, Cout, k, stride, act, use_res, out_f32, use_bias, terms, target = case
    pad = (k - 1) // 2
    g = torch.Generator().manual_seed(_seed(case))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((k, k, Cin, Cout), generator=g) / math.sqrt(k * k * Cin), "w_terms": terms}
    if use_bias:
        s["bias"] = torch.randn((Cout,), generator=g)
    else:
        s["scale"], s["shift"] = torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.1
    if use_res:
        s["residual"] = torch.randn((N, Ho, Wo, Cout), generator=g)
    opts = dict(splitk_target_blocks=target)
    ws = torch.zeros((80 << 20,), dtype=torch.uint8, device=cuda)
    want = _conv_ref(s, k, stride, pad, act, out_f32)
    expect = Cin * k * k * terms // (64 if Cin % 64 == 0 else 32) >= 8     # two parts of four K steps at least
    runs = [_conv_gpu(cuda, [s], k, stride, pad, act, out_f32, opts, splitk_ws=ws, expect_split=expect)[0] for _ in range(3)]
    _close(runs[0], want, out_f32)
    for r in runs[1:]:
        assert torch.equal(r, runs[0])
    assert int(ws[:16384].view(torch.int32).abs().sum().item()) == 0
    whole = _conv_gpu(cuda, [s], k, stride, pad, act, out_f32, opts)[0]
    _close(whole, want, out_f32)
    if not expect:
        assert torch.equal(whole, runs[0])


def test_conv_128_row_split_k_grouped_levels(cuda):
    """A grouped launch (three pyramid levels of a shared 1x1 conv, different depths: the FPN laterals at batch 1) split
    along K: the part count follows the shallowest segment, every segment's tiles exchange through their own slots."""
    g = torch.Generator().manual_seed(77)
    segs = []
    for (H, Cin) in ((20, 2048), (40, 1024), (10, 512)):
        segs.append({"x": torch.randn((1, H, H, Cin), generator=g), "w": torch.randn((1, 1, Cin, 256), generator=g) / math.sqrt(Cin),
                     "bias": torch.randn((256,), generator=g)})
    ws = torch.zeros((80 << 20,), dtype=torch.uint8, device=cuda)
    got = _conv_gpu(cuda, segs, 1, 1, 0, None, False, None, splitk_ws=ws, expect_split=True)
    again = _conv_gpu(cuda, segs, 1, 1, 0, None, False, None, splitk_ws=ws, expect_split=True)
    for s, y, y2 in zip(segs, got, again):
        _close(y, _conv_ref(s, 1, 1, 0, None, False), False)
        assert torch.equal(y, y2)
    assert int(ws[:16384].view(torch.int32).abs().sum().item()) == 0


BALANCED_CASES = [
    # N, H, W, Cin, Cout, act, residual, bias, fused statistics   (1x1 / stride 1 on conv_big_kernel, dispatcher's own choice)
    (2, 160, 160, 64, 256, "relu", True, False, False),    # 200 tiles in one round of 256 workgroups -> 256 tiles of 200 rows
    (8, 80, 80, 128, 512, None, False, False, True),       # raw output + fused BatchNorm statistics: two short blocks per tile
    (5, 80, 80, 256, 1024, None, False, True, False),      # 500 tiles = 1.95 rounds -> whole tiles already fit (no balancing)
    (3, 96, 96, 128, 256, "relu6", False, False, True),    # 108 tiles: fewer than the 256-row kernels take -> 128-row kernel
]


@pytest.mark.parametrize("case", BALANCED_CASES, ids=["x".join(str(v) for v in c) for c in BALANCED_CASES])
def test_conv_big_balanced_tiles(cuda, case):
    """conv_big_kernel's balanced tiles (rn_conv.hip: conv_big_balanced_rows): an HBM-bound 1x1 launch whose 256-row tiles
    would leave the last round of the persistent grid mostly idle runs the same number of rounds with tiles of fewer rows.
    Same values as whole tiles (forced with conv_tile = 2); the fused BatchNorm partial sums come as two blocks per tile
    (rn_conv_bn_row_blocks) and add up to the sums over the stored tensor."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, act, use_res, use_bias, stats = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g), "w": torch.randn((1, 1, Cin, Cout), generator=g) / math.sqrt(Cin)}
    if use_bias:
        s["bias"] = torch.randn((Cout,), generator=g)
    elif not stats:
        s["scale"], s["shift"] = torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.1
    if use_res:
        s["residual"] = torch.randn((N, H, W, Cout), generator=g)
    outs = {}
    for tag, opts in (("auto", None), ("whole", dict(conv_tile=2))):
        p = _C.ConvProblem()
        if opts:
            p.opts = _C.LaunchOpts(**opts)
        p.R = p.S = 1
        p.stride_h = p.stride_w = 1
        p.act, p.out_dtype, p.num_segments = _C.ACT_IDS[act], _C.RN_DT_BF16, 1
        x = _bf(s["x"]).to(cuda).contiguous()
        w = s["w"].to(cuda).float().contiguous()
        wp = torch.empty((lib.rn_conv_cout_pad(Cout), 1, 1, lib.rn_conv_cin_pad(Cin)), dtype=H16, device=cuda)
        _C.check(lib.rn_pack_conv_weight(_C.ptr(w), 1, 1, Cin, Cout, lib.rn_conv_cin_pad(Cin), _C.ptr(wp), _C.current_stream()))
        y = torch.empty((N, H, W, Cout), dtype=H16, device=cuda)
        keep = [x, w, wp]
        sg = p.seg[0]
        sg.x, sg.w, sg.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
        for k in ("scale", "shift", "bias"):
            if k in s:
                t = s[k].to(cuda).float().contiguous()
                keep.append(t)
                setattr(sg, k, t.data_ptr())
        if use_res:
            r = _bf(s["residual"]).to(cuda).contiguous()
            keep.append(r)
            sg.residual = r.data_ptr()
        sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout = N, H, W, Cin, Cin, H, W, Cout
        blocks = lib.rn_conv_bn_row_blocks(ctypes.byref(p), 0)
        part = None
        if stats:
            part = torch.full((blocks, 2, Cout), float("nan"), dtype=torch.float32, device=cuda)
            sg.bn_partial = part.data_ptr()
        kid = lib.rn_conv_kernel_id(ctypes.byref(p))
        _C.check(lib.rn_conv2d_nhwc_fwd(ctypes.byref(p), _C.current_stream()), "conv")
        torch.cuda.synchronize()
        outs[tag] = (y.float().cpu(), None if part is None else part.double().cpu(), blocks, kid)
    M = N * H * W
    ya, pa, ba, ka = outs["auto"]
    yw, pw, bw, kw = outs["whole"]
    assert kw == 1 and bw == 2 * ((M + 255) // 256)
    if ka == 1:
        balanced = ba != bw
        assert balanced == (case in BALANCED_CASES[:2]), (ba, bw)
        assert torch.equal(ya, yw)          # a tile's accumulation order does not depend on the rows it covers
    else:
        assert ka == 0 and case == BALANCED_CASES[3]
    _close(ya, _conv_ref(s, 1, 1, 0, act, False), False)
    if stats:
        stored = ya.double().reshape(-1, Cout)
        want = torch.stack([stored.sum(0), (stored * stored).sum(0)])
        for part in (pa, pw):
            assert torch.isfinite(part).all()
            torch.testing.assert_close(part.sum(0), want, rtol=2e-5, atol=2e-5 * want.abs().max().item())


def test_split_k_hand_off_under_uneven_load(cuda):
    """The split-K hand-offs (128-row kernel: per-tile ticket, last arriver reduces; halo kernel: per-wave tickets) use
    write-through stores, a drained vmcnt, one returning agent-scope atomic and sc1 loads — no cache-wide release / acquire
    (cdna_hip_programming.md guideline 16, R1).  Such a protocol has to be tested where it can fail: uneven load (a second
    stream streams HBM and keeps part of the chip busy, so the parts of a tile arrive far apart), slots that are L2 / L1
    warm from the previous launch, every output word checked.  300 launches of each form, alternating two different
    inputs on the same workspace: every result must equal the first result of its input bit for bit."""
    from retinanet import _C
    lib = _lib()
    g = torch.Generator().manual_seed(99)
    cases = [(1, 20, 20, 512, 512, 3, dict()),                     # 128-row kernel, 8 parts per tile
             (1, 30, 30, 512, 256, 3, dict(conv_tile=2)),          # halo kernel, 4 tiles x 4 parts, per-wave tickets
             (1, 20, 20, 2048, 512, 1, dict())]                    # 128-row kernel, 1x1, 8 parts of 4 K steps
    side = torch.cuda.Stream(cuda)
    noise_a = torch.randn((64 << 20,), device=cuda)
    noise_b = torch.empty_like(noise_a)
    for (N, H, W, Cin, Cout, k, opts) in cases:
        ws = torch.zeros((80 << 20,), dtype=torch.uint8, device=cuda)
        inputs = []
        for _ in range(2):
            s = {"x": torch.randn((N, H, W, Cin), generator=g), "w": torch.randn((k, k, Cin, Cout), generator=g) / math.sqrt(k * k * Cin),
                 "scale": torch.rand((Cout,), generator=g) + 0.5, "shift": torch.randn((Cout,), generator=g) * 0.1}
            inputs.append(s)
        # build both problems once (device tensors stay alive), then alternate launches
        built = []
        for s in inputs:
            p = _C.ConvProblem()
            p.opts = _C.LaunchOpts(**opts)
            p.R = p.S = k
            p.stride_h = p.stride_w = 1
            p.pad_top = p.pad_left = (k - 1) // 2
            p.act, p.out_dtype, p.num_segments = _C.ACT_IDS["relu"], _C.RN_DT_BF16, 1
            x = _bf(s["x"]).to(cuda).contiguous()
            w = s["w"].to(cuda).float().contiguous()
            cinp = lib.rn_conv_cin_pad(Cin)
            wp = torch.empty((lib.rn_conv_cout_pad(Cout), k, k, cinp), dtype=H16, device=cuda)
            _C.check(lib.rn_pack_conv_weight(_C.ptr(w), k, k, Cin, Cout, cinp, _C.ptr(wp), _C.current_stream()))
            y = torch.empty((N, H, W, Cout), dtype=H16, device=cuda)
            sc, sh = s["scale"].to(cuda), s["shift"].to(cuda)
            sg = p.seg[0]
            sg.x, sg.w, sg.y, sg.scale, sg.shift = x.data_ptr(), wp.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr()
            sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout = N, H, W, Cin, Cin, H, W, Cout
            p.splitk_ws, p.splitk_ws_bytes = ws.data_ptr(), ws.numel()
            assert lib.rn_conv_splitk_workspace_bytes(ctypes.byref(p)) > 0
            built.append((p, y, [x, w, wp, sc, sh]))
        torch.cuda.synchronize()
        first = [None, None]
        st = _C.current_stream()
        for it in range(300):
            if it % 3 == 0:
                with torch.cuda.stream(side):                      # uneven load: 256 MB streamed beside the launches
                    torch.add(noise_a, 1.0, out=noise_b)
            j = it & 1
            p, y, _ = built[j]
            _C.check(lib.rn_conv2d_nhwc_fwd(ctypes.byref(p), st), "conv")
            if it < 2 or it % 25 == 0 or it >= 296:
                got = y.clone()
                torch.cuda.synchronize()
                if first[j] is None:
                    first[j] = got
                else:
                    assert torch.equal(got, first[j]), (k, Cin, Cout, it)
        torch.cuda.synchronize()
        for j in (0, 1):
            assert torch.equal(built[j][1], first[j])
            _close(first[j].float().cpu(), _conv_ref(inputs[j], k, 1, (k - 1) // 2, "relu", False), False)
        assert int(ws[:16384].view(torch.int32).abs().sum().item()) == 0


HALO512_CASES = [
    # N, H, W, Cin, Cout, act, residual, out_f32, persistent workgroups   (3x3 / stride 1 / pad 1, 64 < Cout <= 128)
    (2, 80, 80, 128, 128, "relu", False, False, 0),    # ResNet stage 2 at its own width: 6.4 rows per 512-pixel tile, 4 chunks
    (3, 24, 24, 128, 128, "relu", True, False, 2),     # tiles straddle images; 4 tiles over 2 workgroups (two rounds)
    (5, 5, 5, 64, 96, None, False, False, 0),          # one tile over five tiny images (many pad rows); Cout tail (96 of 128)
    (1, 40, 40, 256, 128, "relu6", False, False, 3),   # 8 chunks, M tail (1600 = 3 tiles + 64 pixels)
    (2, 13, 9, 96, 72, "swish", True, False, 0),       # odd sizes, 3 chunks, Cout = 72
    (1, 20, 20, 64, 128, None, False, True, 0),        # f32 output
    (40, 3, 3, 32, 128, None, False, False, 0),        # single chunk; 56 images per tile
    (2, 1, 7, 64, 128, None, False, False, 0),         # H = 1
]


@pytest.mark.parametrize("build,case", _builds(HALO512_CASES, HALO512_CASES[:5], "halo512-"))
def test_conv_halo_kernel_512x128_tiles(cuda, build, case):
    """3x3 / stride 1 / pad 1 with 64 < Cout <= 128 through the halo kernel's 512 x 128 form (rn_conv_halo.hip, HaloGeo<4>:
    4 x 2 waves, 13-block patches, one weight piece per wave), against the float64 reference and against the 128-row
    kernel on the same inputs (same products, other summation order)."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, act, use_res, out_f32, wgs = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((3, 3, Cin, Cout), generator=g) / math.sqrt(9 * Cin),
         "scale": torch.rand((Cout,), generator=g) + 0.5, "shift": torch.randn((Cout,), generator=g) * 0.1}
    if use_res:
        s["residual"] = torch.randn((N, H, W, Cout), generator=g)
    p = _C.ConvProblem()
    p.R = p.S = 3
    p.stride_h = p.stride_w = p.pad_top = p.pad_left = 1
    p.out_dtype, p.num_segments = (_C.RN_DT_F32 if out_f32 else _C.RN_DT_BF16), 1
    sg = p.seg[0]
    sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout = N, H, W, Cin, Cin, H, W, Cout
    p.opts = _C.LaunchOpts(conv_tile=2)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == 3 and lib.rn_conv_tile_rows(ctypes.byref(p)) == 512
    got = _conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, dict(conv_tile=2, max_workgroups=wgs))[0]
    p.opts = _C.LaunchOpts(conv_tile=1)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == 0
    small = _conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, dict(conv_tile=1))[0]
    want = _conv_ref(s, 3, 1, 1, act, out_f32)
    _close(got, want, out_f32)
    scale = want.abs().max().item() + 1e-6
    torch.testing.assert_close(got, small, rtol=1.0 / 128 if not out_f32 else 1e-4, atol=scale * (1 / 256 if not out_f32 else 1e-5))
    if not out_f32:
        assert (got != small).float().mean().item() < 0.02


HALO512_WIDE_CASES = [
    # N, H, W, Cin, Cout, act, residual, f32 output, bias (instead of scale / shift), w_terms, persistent workgroups
    (2, 40, 40, 64, 256, "relu", False, False, True, 1, 0),     # head tower: two column tiles per pixel tile, bias + relu
    (1, 33, 31, 96, 256, "relu", True, False, False, 1, 3),     # folded BatchNorm + residual + relu, 3 workgroups (4 rounds)
    (2, 20, 20, 64, 720, None, False, True, True, 2, 0),        # class prediction: f32, two weight planes along Cin, 6 column tiles (the last 80 channels live)
    (3, 16, 16, 128, 384, None, False, False, False, 1, 2),     # three column tiles, raw output
    (2, 24, 24, 64, 320, "relu6", False, False, True, 1, 0),    # Cout tail: the third column tile has 64 live channels
]


@pytest.mark.parametrize("build,case", _builds(HALO512_WIDE_CASES, HALO512_WIDE_CASES[:3], "halo512w-"))
def test_conv_halo_kernel_512x128_tiles_wide_layers(cuda, build, case):
    """The 512 x 128 form on layers of 256 and more channels (what the dispatcher picks for the head towers, the FPN output
    convs and the class prediction conv from three rounds of tiles on; forced here with conv_tile = 3): several column tiles
    per pixel tile, against the float64 reference and against the 256 x 256 form on the same inputs."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, act, use_res, out_f32, use_bias, terms, wgs = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((3, 3, Cin, Cout), generator=g) / math.sqrt(9 * Cin), "w_terms": terms}
    if use_bias:
        s["bias"] = torch.randn((Cout,), generator=g) * 0.5
    elif act is not None:
        s["scale"], s["shift"] = torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.1
    if use_res:
        s["residual"] = torch.randn((N, H, W, Cout), generator=g)
    p = _C.ConvProblem()
    p.R = p.S = 3
    p.stride_h = p.stride_w = p.pad_top = p.pad_left = 1
    p.out_dtype, p.num_segments = (_C.RN_DT_F32 if out_f32 else _C.RN_DT_BF16), 1
    sg = p.seg[0]
    sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout = N, H, W, Cin, Cin, H, W, Cout
    sg.w_terms = terms
    p.opts = _C.LaunchOpts(conv_tile=3)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == 3 and lib.rn_conv_tile_rows(ctypes.byref(p)) == 512
    got = _conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, dict(conv_tile=3, max_workgroups=wgs))[0]
    p.opts = _C.LaunchOpts(conv_tile=2)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == 2
    wide = _conv_gpu(cuda, [s], 3, 1, 1, act, out_f32, dict(conv_tile=2))[0]
    want = _conv_ref(s, 3, 1, 1, act, out_f32)
    _close(got, want, out_f32)
    # the same products in the same K order per accumulator: the two tile shapes agree to the bit
    assert torch.equal(got, wide)


HALO512_NARROW_CASES = [
    # N, H, W, Cin, Cout, act, residual
    (2, 40, 40, 64, 64, "relu", False),      # ResNet stage 1 3x3 at its own width: half of the tile's columns are zero weights
    (1, 33, 31, 64, 40, None, False),        # Cout tail inside the live half
    (3, 16, 16, 32, 64, "relu", True),       # one chunk, residual
]


@pytest.mark.parametrize("case", HALO512_NARROW_CASES, ids=lambda c: "halo512n-" + "x".join(str(v) for v in c))
def test_conv_halo_kernel_512x128_tiles_narrow_layers(cuda, case):
    """Cout <= 64 on the 512 x 128 form (conv_tile = 3): the packed weights have 64 rows, the tile's other 64 columns read
    zeros through the buffer bounds.  Against the float64 reference and the 128-row kernel."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, act, use_res = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((3, 3, Cin, Cout), generator=g) / math.sqrt(9 * Cin),
         "scale": torch.rand((Cout,), generator=g) + 0.5, "shift": torch.randn((Cout,), generator=g) * 0.1}
    if use_res:
        s["residual"] = torch.randn((N, H, W, Cout), generator=g)
    p = _C.ConvProblem()
    p.R = p.S = 3
    p.stride_h = p.stride_w = p.pad_top = p.pad_left = 1
    p.out_dtype, p.num_segments = _C.RN_DT_BF16, 1
    sg = p.seg[0]
    sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout = N, H, W, Cin, Cin, H, W, Cout
    p.opts = _C.LaunchOpts(conv_tile=3)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == 3 and lib.rn_conv_tile_rows(ctypes.byref(p)) == 512
    got = _conv_gpu(cuda, [s], 3, 1, 1, act, False, dict(conv_tile=3))[0]
    small = _conv_gpu(cuda, [s], 3, 1, 1, act, False, dict(conv_tile=1))[0]
    want = _conv_ref(s, 3, 1, 1, act, False)
    _close(got, want, False)
    scale = want.abs().max().item() + 1e-6
    torch.testing.assert_close(got, small, rtol=1.0 / 128, atol=scale / 256)


BIAS_CASES = [
    # N, H, W, Cin, Cout, k, act, persistent workgroups (0 = one per CU)
    (2, 24, 24, 64, 256, 3, "relu", 0),
    (2, 24, 24, 64, 256, 3, "relu", 2),      # 5 tiles over 2 workgroups: accumulators re-initialised in the loop
    (7, 10, 10, 128, 320, 3, "relu6", 3),    # Cout tail: the second n-tile has 64 live channels
    (1, 20, 20, 64, 720, 3, None, 2),        # three n-tiles, the last with 208 live channels
    (3, 17, 19, 128, 512, 1, "relu", 3),     # 1x1 on conv_big_kernel, two n-tiles, M tail
    (1, 20, 20, 256, 264, 1, None, 0),       # Cout = 264: one live 8-channel group in the second n-tile
]


@pytest.mark.parametrize("case", BIAS_CASES, ids=lambda c: "bias-" + "x".join(str(v) for v in c))
def test_conv_bias_only_rounds_once(cuda, case):
    """Conv2D + bias on the 256-row kernels: the accumulators start at the bias (rn_conv_big_epi.h big_acc_init), so
    the layer's output is bf16(act(sum + bias)) with ONE rounding (fp32 BiasAdd inside the layer, then the cast):
    against the float64 restatement at most a sliver of the outputs may sit one bf16 step off (fp32 summation
    order)."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, k, act, grid = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((k, k, Cin, Cout), generator=g) / math.sqrt(k * k * Cin),
         "bias": torch.randn((Cout,), generator=g)}
    pad = (k - 1) // 2
    got = _conv_gpu(cuda, [s], k, 1, pad, act, False, dict(conv_tile=2, max_workgroups=grid))[0]
    want = _conv_ref(s, k, 1, pad, act, False)
    _close(got, want, False)
    assert (got != want).float().mean().item() < 2e-3


LAYER_CASES = [
    # N, H, W, Cin, Cout, k, act, residual, kernel (0 = 128-row, 1 = conv_big, 2 = conv_halo)
    (2, 20, 20, 256, 256, 3, None, False, 0),     # FPN output conv: Conv2D + bias -> bf16 -> BatchNorm -> bf16
    (2, 20, 20, 256, 256, 3, None, False, 2),
    (2, 20, 20, 512, 256, 1, None, False, 1),     # FPN lateral 1x1 + bias + BN
    (2, 20, 20, 256, 256, 3, "relu", False, 2),   # head tower at inference: conv + bias, BN, relu
    (2, 12, 12, 64, 256, 1, "relu", True, 0),     # bias AND residual: stays on the 128-row kernel
]


@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: "layer-" + "x".join(str(v) for v in c))
def test_conv_bias_then_batchnorm_rounding_points(cuda, case):
    """Conv2D(+bias) -> BatchNormalization (-> + residual) as ONE launch: bias before the first rounding, scale /
    shift after it, the residual after the second — on every kernel the dispatcher can pick."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, k, act, use_res, kid = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g),
         "w": torch.randn((k, k, Cin, Cout), generator=g) / math.sqrt(k * k * Cin),
         "bias": torch.randn((Cout,), generator=g),
         "scale": torch.rand((Cout,), generator=g) + 0.5, "shift": torch.randn((Cout,), generator=g) * 0.1}
    if use_res:
        s["residual"] = torch.randn((N, H, W, Cout), generator=g)
    pad = (k - 1) // 2
    got = _conv_gpu(cuda, [s], k, 1, pad, act, False, dict(conv_tile=2 if kid else 1))[0]
    _close(got, _conv_ref(s, k, 1, pad, act, False), False)


SPLIT_CASES = [
    # N, H, W, Cin, Cout, k, terms, kernel (0 = 128-row, 1 = conv_big, 2 = conv_halo)
    (2, 5, 5, 256, 36, 3, 2, 0),         # box prediction conv
    (1, 10, 10, 256, 720, 3, 2, 0),      # class prediction conv on the 128-row kernel (small launch)
    (1, 20, 20, 256, 720, 3, 2, 2),      # ... on the halo kernel (what batch 8 / 32 at 640 x 640 runs)
    (1, 20, 20, 256, 720, 3, 3, 2),
    (1, 20, 20, 256, 720, 3, 2, 1),      # ... on conv_big_kernel (halo switched off)
    (2, 16, 16, 160, 720, 1, 2, 1),      # pointwise half of a separable prediction conv (EfficientNet-B3 widths)
    (2, 16, 16, 160, 36, 1, 3, 0),
]


@pytest.mark.parametrize("build,case", _builds(SPLIT_CASES, SPLIT_CASES[:4], "split-"))
def test_conv_f32_weights_as_split_bf16_planes(cuda, build, case):
    """The dtype=float32 prediction convs (detection_head.py:80-88): bf16 activations x f32 kernel + f32 bias,
    f32 accumulate, f32 out.  With rn_conv_segment.w_terms the kernel is carried as 2 (3) bf16 planes: against the
    float64 product with the SAME planes only fp32 summation order is left (<= 2e-5 of the output range); against
    the product with the untouched f32 kernel the 2-plane form is within 2^-16 per weight (<= 3e-5), the 3-plane
    form exact to fp32.  One plane (plain bf16 weights, what round 1 shipped) is ~1e-3 off: measured below."""
    from retinanet import _C
    lib = _lib()
    N, H, W, Cin, Cout, k, terms, kid = case
    g = torch.Generator().manual_seed(_seed(case))
    s = {"x": torch.randn((N, H, W, Cin), generator=g).relu(),          # tower outputs are post-ReLU
         "w": torch.randn((k, k, Cin, Cout), generator=g) * 0.01,      # RandomNormal(0.01), detection_head.py:40-43
         "bias": torch.full((Cout,), -4.59512), "w_terms": terms}
    pad = (k - 1) // 2
    opts = dict(conv_tile=2 if kid else 1, conv_no_halo=0 if kid == 2 else 1)
    p = _C.ConvProblem()
    p.R = p.S = k
    p.stride_h = p.stride_w = 1
    p.pad_top = p.pad_left = pad
    p.out_dtype, p.num_segments = _C.RN_DT_F32, 1
    sg = p.seg[0]
    sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout = N, H, W, Cin, Cin, H, W, Cout
    p.opts = _C.LaunchOpts(**opts)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == kid
    got = _conv_gpu(cuda, [s], k, 1, pad, None, True, opts)[0]
    one = _conv_gpu(cuda, [dict(s, w_terms=1)], k, 1, pad, None, True, opts)[0]
    same_planes = _conv_ref(s, k, 1, pad, None, True)
    exact = _conv_ref(dict(s, w_terms=3), k, 1, pad, None, True)
    spread = (exact + 4.59512).abs().max().item()       # range of the logits around the bias
    assert (got - same_planes).abs().max().item() <= 2e-5 * spread
    assert (got - exact).abs().max().item() <= 3e-5 * spread
    # what the split buys: plain bf16 weights are two orders of magnitude further from the f32 layer
    assert (one - exact).abs().max().item() > 10 * (got - exact).abs().max().item()


PAIR_CASES = [
    # per segment (N, H, W); Cin, Cout, k, conv_tile, no_halo -> kernel id
    ([(2, 24, 24), (2, 12, 12), (3, 5, 5)], 256, 36, 3, 2, 0, 3),    # the box prediction conv: 512 x 128 halo tiles, 3 levels
    ([(1, 33, 17)], 64, 4, 3, 2, 0, 3),                              # one 4-channel group; ragged rows
    ([(2, 20, 20)], 96, 64, 3, 2, 0, 3),                             # both 32-channel blocks full
    ([(1, 16, 16)], 128, 100, 3, 2, 0, 2),                           # 256 GEMM columns: the 256 x 256 halo tiles
    ([(1, 16, 16)], 128, 132, 1, 2, 0, 1),                           # 1x1: conv_big_kernel, 384 columns (the last block half used)
]


@pytest.mark.parametrize("build,case", _builds(PAIR_CASES, PAIR_CASES[:2], "pair-"))
def test_conv_f32_weight_planes_along_cout(cuda, build, case):
    """rn_conv_segment.w_pair: the two split-bf16 planes of an f32 kernel as GEMM columns (hi | lo per 32-channel block),
    added in the epilogue.  Same bar as the planes-along-Cin form (test_conv_f32_weights_as_split_bf16_planes): against the
    float64 product with the same planes only fp32 summation order is left; against the untouched f32 kernel 2^-16 per
    weight.  Channel counts that are multiples of 4 only (36 box-regression channels), several pyramid levels per launch,
    and a launch the 128-row kernel would get is refused."""
    from retinanet import _C
    lib = _lib()
    shapes, Cin, Cout, k, tile, no_halo, kid = case
    g = torch.Generator().manual_seed(_seed(case))
    segs = [{"x": torch.randn((N, H, W, Cin), generator=g).relu(),
             "w": torch.randn((k, k, Cin, Cout), generator=g) * 0.01,
             "bias": torch.randn((Cout,), generator=g) * 0.1, "w_pair": 1} for (N, H, W) in shapes]
    for s in segs[1:]:
        s["w"], s["bias"] = segs[0]["w"], segs[0]["bias"]             # one kernel shared by the levels
    pad = (k - 1) // 2
    opts = dict(conv_tile=tile, conv_no_halo=no_halo)
    p = _C.ConvProblem()
    p.R = p.S = k
    p.stride_h = p.stride_w = 1
    p.pad_top = p.pad_left = pad
    p.out_dtype, p.num_segments = _C.RN_DT_F32, len(shapes)
    for i, (N, H, W) in enumerate(shapes):
        sg = p.seg[i]
        sg.N, sg.H, sg.W, sg.Cin, sg.pix_stride, sg.Ho, sg.Wo, sg.Cout, sg.w_pair = N, H, W, Cin, Cin, H, W, Cout, 1
    p.opts = _C.LaunchOpts(**opts)
    assert lib.rn_conv_kernel_id(ctypes.byref(p)) == kid
    assert lib.rn_conv_pair_rows(Cout) == 128 * ((Cout + 63) // 64)
    gots = _conv_gpu(cuda, segs, k, 1, pad, None, True, opts)
    for s, got in zip(segs, gots):
        same_planes = _conv_ref(dict(s, w_terms=2), k, 1, pad, None, True)
        exact = _conv_ref(dict(s, w_terms=3), k, 1, pad, None, True)
        spread = (exact - s["bias"].double()).abs().max().item() + 1e-9
        assert torch.isfinite(got).all()
        assert (got - same_planes).abs().max().item() <= 2e-5 * spread
        assert (got - exact).abs().max().item() <= 3e-5 * spread
    # the 128-row kernel has no such epilogue: refused, not silently wrong
    with pytest.raises(_C.RnetError):
        _conv_gpu(cuda, segs[:1], k, 1, pad, None, True, dict(conv_tile=1))


def test_conv_halo_asymmetric_weights(cuda):
    """Exact shift test through the halo kernel: output channel c = input channel (c+1)%C read through tap
    (r=0, s=2), i.e. pixel (y-1, x+1); borders must be exact zeros."""
    from retinanet import _C
    lib = _lib()
    C = 256
    x = (torch.arange(2 * 9 * 11 * C, dtype=torch.float32).reshape(2, 9, 11, C) * 7) % 61
    w = torch.zeros((3, 3, C, C))
    for c in range(C):
        w[0, 2, (c + 1) % C, c] = 1.0
    got = _conv_gpu(cuda, [{"x": x, "w": w}], 3, 1, 1, None, True, dict(conv_tile=2))[0]
    want = torch.zeros_like(got)
    want[:, 1:, :-1, :] = torch.roll(x, -1, dims=3)[:, :-1, 1:, :]
    torch.testing.assert_close(got, want, rtol=0, atol=0)


def test_conv_asymmetric_weights_detect_transposes(cuda):
    """A=I style check with an asymmetric filter: channel c of the output must be input channel
    (c+1)%C shifted by one pixel — catches row/col or r/s swaps that random data would blur."""
    C = 64
    x = torch.arange(1 * 6 * 6 * C, dtype=torch.float32).reshape(1, 6, 6, C) % 97
    w = torch.zeros((3, 3, C, C))
    for c in range(C):
        w[0, 2, (c + 1) % C, c] = 1.0      # tap (r=0, s=2): reads pixel (y-1, x+1)
    got = _conv_gpu(cuda, [{"x": x, "w": w}], 3, 1, 1, None, True)[0]
    want = torch.zeros_like(got)
    want[:, 1:, :-1, :] = torch.roll(x, -1, dims=3)[:, :-1, 1:, :]
    torch.testing.assert_close(got, want, rtol=0, atol=0)


def _run_grouped_pyramid(cuda, opts=None):
    """One launch, 10 segments: the two head towers over five pyramid levels (shared kernel
    size, per-segment weights/BN) — the shape of detection_head.py:94-100."""
    g = torch.Generator().manual_seed(5)
    segs = []
    for head in range(2):
        w = torch.randn((3, 3, 256, 256), generator=g) / 48.0
        for s in (16, 8, 4, 2, 1):
            segs.append({"x": torch.randn((2, s, s, 256), generator=g), "w": w,
                         "scale": torch.rand((256,), generator=g) + 0.5, "shift": torch.randn((256,), generator=g) * 0.1})
    got = _conv_gpu(cuda, segs, 3, 1, 1, "relu", False, opts)
    for s, y in zip(segs, got):
        _close(y, _conv_ref(s, 3, 1, 1, "relu", False), False)


def test_conv_grouped_pyramid(cuda):
    _run_grouped_pyramid(cuda)


def test_stem_conv(cuda):
    """7x7 s2 stem through the packed NHWC4 image + [64][7][32] weights (resnet.py:297-300)."""
    from retinanet import _C
    lib = _lib()
    g = torch.Generator().manual_seed(9)
    N, H, W = 2, 64, 96
    img = torch.randn((N, H, W, 3), generator=g)
    w = torch.randn((7, 7, 3, 64), generator=g) / math.sqrt(147)
    scale, shift = torch.rand((64,), generator=g) + 0.5, torch.randn((64,), generator=g) * 0.1
    Wp = lib.rn_stem_padded_width(W)
    imd = img.to(cuda).contiguous()
    packed = torch.empty((N, H + 6, Wp, 4), dtype=H16, device=cuda)
    _C.check(lib.rn_pack_stem_input(_C.ptr(imd), N, H, W, _C.ptr(packed), _C.current_stream()))
    wd = w.to(cuda).contiguous()
    wp = torch.empty((64, 7, 32), dtype=H16, device=cuda)
    _C.check(lib.rn_pack_stem_weight(_C.ptr(wd), 64, _C.ptr(wp), _C.current_stream()))
    y = torch.empty((N, H // 2, W // 2, 64), dtype=H16, device=cuda)
    sc, sh = scale.to(cuda), shift.to(cuda)
    p = _C.ConvProblem()
    p.R, p.S, p.stride_h, p.stride_w, p.pad_top, p.pad_left = 7, 1, 2, 2, 0, 0
    p.act, p.out_dtype, p.num_segments = _C.RN_ACT_RELU, _C.RN_DT_BF16, 1
    s = p.seg[0]
    s.x, s.w, s.y, s.scale, s.shift, s.residual = packed.data_ptr(), wp.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), None
    s.N, s.H, s.W, s.Cin, s.pix_stride, s.Ho, s.Wo, s.Cout = N, H + 6, Wp, 32, 4, H // 2, W // 2, 64
    _C.check(lib.rn_conv2d_nhwc_fwd(ctypes.byref(p), _C.current_stream()))
    torch.cuda.synchronize()
    # packed image: border and 4th channel zero, interior = bf16(image)
    pk = packed.float().cpu()
    assert (pk[:, :3] == 0).all() and (pk[:, :, :3] == 0).all() and (pk[..., 3] == 0).all()
    torch.testing.assert_close(pk[:, 3:3 + H, 3:3 + W, :3], _bf(img).float(), rtol=0, atol=0)
    want = _conv_ref({"x": img, "w": w, "scale": scale, "shift": shift}, 7, 2, 3, "relu", False)
    _close(y.float().cpu(), want, False)


def test_maxpool_same_and_valid(cuda):
    from retinanet import _C
    lib = _lib()
    g = torch.Generator().manual_seed(2)
    x = _bf(torch.randn((2, 20, 20, 64), generator=g))
    xd = x.to(cuda)
    y = torch.empty((2, 10, 10, 64), dtype=H16, device=cuda)
    # 3x3 s2 SAME on an even size: pad 0 top/left, 1 bottom/right
    _C.check(lib.rn_maxpool2d_nhwc(_C.ptr(xd), _C.ptr(y), 2, 20, 20, 64, 3, 2, 0, 0, 10, 10, _C.current_stream()))
    xp = F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1), value=float("-inf"))
    torch.testing.assert_close(y.float().cpu(), F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1), rtol=0, atol=0)
    _C.check(lib.rn_maxpool2d_nhwc(_C.ptr(xd), _C.ptr(y), 2, 20, 20, 64, 2, 2, 0, 0, 10, 10, _C.current_stream()))
    torch.testing.assert_close(y.float().cpu(), F.max_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1), rtol=0, atol=0)


def _pyramid(g, N, H0, C, L=5):
    return [_bf(torch.randn((N, H0 >> l, H0 >> l, C), generator=g)) for l in range(L)]


@pytest.mark.parametrize("act,N,H0,C,L", [("relu", 2, 32, 64, 5), ("relu6", 2, 32, 64, 5),
                                          ("relu", 16, 80, 256, 5),    # >= 2^21 16-byte items on the finest level: the
                                          ("relu6", 24, 64, 256, 3)])  # pyramid is cut into one-stage launches
def test_fpn_topdown(cuda, act, N, H0, C, L):
    from retinanet import _C
    lib = _lib()
    g = torch.Generator().manual_seed(3)
    pyr = _pyramid(g, N, H0, C, L)
    ins = [p.to(cuda) for p in pyr]
    outs = [torch.empty_like(t) for t in ins[:-1]] + [ins[-1]]
    _C.check(lib.rn_fpn_topdown(_C.ptr_array(ins), _C.ptr_array(outs), L, N, H0, H0, C, _C.ACT_IDS[act], _C.current_stream()))
    torch.cuda.synchronize()
    ref = [p.float() for p in pyr]
    for l in range(L - 1, 0, -1):   # fpn.py:93-98
        up = F.interpolate(ref[l].permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
        v = ref[l - 1] + up
        v = F.relu(v) if act == "relu" else F.relu6(v)
        ref[l - 1] = _bf(v).float()
    for l in range(L):
        assert torch.equal(outs[l].float().cpu(), ref[l]), l


def test_balance_features(cuda):
    from retinanet import _C
    lib = _lib()
    g = torch.Generator().manual_seed(4)
    pyr = _pyramid(g, 2, 32, 64)
    ts = [p.to(cuda).clone() for p in pyr]
    scratch = torch.empty_like(ts[1])
    arr = _C.ptr_array(ts)
    _C.check(lib.rn_balance_features(arr, arr, 5, 1, 2, 32, 32, 64, _C.ptr(scratch), _C.current_stream()))
    torch.cuda.synchronize()
    f = [p.float().permute(0, 3, 1, 2) for p in pyr]
    rs = [F.max_pool2d(f[0], 2), f[1]] + [F.interpolate(f[l], scale_factor=2 ** (l - 1), mode="nearest") for l in (2, 3, 4)]
    avg = rs[0]
    for r in rs[1:]:
        avg = avg + r
    avg = _bf(avg / 5.0).float()
    back = [F.interpolate(avg, scale_factor=2, mode="nearest"), avg] + [F.max_pool2d(avg, 2 ** (l - 1)) for l in (2, 3, 4)]
    for l in range(5):
        want = _bf(f[l] + back[l]).float().permute(0, 2, 3, 1)
        torch.testing.assert_close(ts[l].float().cpu(), want, rtol=1 / 128, atol=1e-2)
