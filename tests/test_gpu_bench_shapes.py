"""Per-launch parity AT THE BENCH'S EXACT SHAPES.

The whole-network training tests bound the HIP path by the restatement's own arithmetic noise (gradient cosines around
0.93 at ResNet-50): that proves wiring, not arithmetic.  The sharp evidence is per kernel — and the kernel tests of
tests/test_gpu_conv.py / tests/test_gpu_train_kernels.py run at toy shapes (a handful of tiles).  Here every DISTINCT
implicit-GEMM launch (forward, data gradient, weight gradient: kernel id x geometry x epilogue inputs) that the
training engine of bench.py issues — B = 32 at 640 x 640 (BASELINE configs[2] shard), and the 1024 x 1024 engine
(configs[3]) — is re-issued through the C ABI on random 16-bit inputs with the launch's own descriptor geometry, and
compared with a float64 evaluation of the same layer (reference semantics: Conv2D of resnet.py:118-144 /
detection_head.py:56-88 and its autodiff) that rounds where include/rnet_hip.h rn_conv_segment rounds.  This covers
what a toy shape cannot: multi-round persistent grids (up to 11 tiles per CU), the XCD-aware tile walk over > 2000
tiles, split-K plans of the weight-gradient kernels at full size, the float-reciprocal index arithmetic near its
2^22-pixel bound, 2 GiB buffer-addressing limits.

The float64 reference runs on the GPU as per-tap matrix products (torch / rocBLAS dgemm — an independent code path):
  forward / dgrad :  y[p, :] = sum over taps  x[p + tap, :] @ w[tap]
  weight gradient :  dw[:, tap, :] = dy[p, :]^T @ x[p + tap, :]
Criteria: bf16 outputs as tests/test_gpu_conv.py::_close (shared rounding points: one bf16 step where an fp32 sum
straddles a rounding boundary, on < 3 % of the outputs); f32 outputs (split-plane prediction convs) to 2e-4 of the
range; weight gradients to 1e-3 of the tensor's largest entry; fused BatchNorm partial sums to fp32 summation accuracy.
"""
import ctypes
import math
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _engine(cuda, size, B):
    from test_gpu_train_step import _setup
    p, model, eng, targets, images = _setup(cuda, size, B, True, depth=50, freeze=True)
    return eng


@pytest.fixture
def splitk_env(request, monkeypatch):
    """'split': the engine attaches a split-K workspace to its conv launches (RNET_SPLITK=1: off by default, it does not
    pay — retinanet/_C.py::new_splitk_workspace), so the launches checked below run their last round split along K"""
    if request.param == "split":
        monkeypatch.setenv("RNET_SPLITK", "1")
    return request.param


def _conv_sig(eng, name, p):
    segs = tuple((s.N, s.H, s.W, s.Cin, s.pix_stride, s.Ho, s.Wo, s.Cout, bool(s.scale), bool(s.shift), bool(s.bias),
                  bool(s.residual), bool(s.residual) and s.residual == s.y, s.w_terms, s.w_pair, bool(s.bn_partial),
                  bool(s.bn_bwd_y)) for s in (p.seg[i] for i in range(p.num_segments)))
    return (p.R, p.S, p.stride_h, p.pad_top, p.act, p.out_dtype, eng.lib.rn_conv_kernel_id(ctypes.byref(p)), segs)


def _wgrad_sig(name, p):
    segs = tuple((s.N, s.H, s.W, s.Cin, s.Ho, s.Wo, s.Cout, s.dy_pix_stride, s.x_pix_stride)
                 for s in (p.seg[i] for i in range(p.num_segments)))
    return (p.R, p.S, p.stride_h, p.pad_top, segs)


def _rb(t, h16):
    """round a float64 tensor where the kernels hold a 16-bit tensor (through fp32, like the fp32 accumulators)"""
    return t.float().to(h16).double()


def _tap_views(xp, R, S, stride, Ho, Wo):
    for r in range(R):
        for s in range(S):
            yield r, s, xp[:, r:r + stride * (Ho - 1) + 1:stride, s:s + stride * (Wo - 1) + 1:stride, :]


def _pad_input(x, R, S, stride, pt, pl, Ho, Wo):
    N, H, W, C = x.shape
    pb = max((Ho - 1) * stride + R - H - pt, 0)
    pr = max((Wo - 1) * stride + S - W - pl, 0)
    return F.pad(x, (0, 0, pl, pr, pt, pb))


def _close_bf16(got, want):
    scale = want.abs().max().item() + 1e-6
    torch.testing.assert_close(got, want, rtol=1.0 / 128, atol=scale / 256)
    assert (got - want).abs().mean().item() <= 2e-3 * scale
    assert (got != want).float().mean().item() < 0.03


def _check_conv_launch(cuda, eng, name, p):
    """re-issue one forward / data-gradient launch on fresh random tensors of its own geometry"""
    from retinanet import _C
    lib, h16 = eng.lib, eng.h16
    g = torch.Generator(device=cuda).manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
    q = _C.ConvProblem()
    q.R, q.S, q.stride_h, q.stride_w, q.pad_top, q.pad_left = p.R, p.S, p.stride_h, p.stride_w, p.pad_top, p.pad_left
    q.act, q.out_dtype, q.num_segments, q.opts = p.act, p.out_dtype, p.num_segments, p.opts
    f32 = p.out_dtype == _C.RN_DT_F32
    rows = lib.rn_conv_tile_rows(ctypes.byref(p))
    keep, per_seg = [], []
    for i in range(p.num_segments):
        s, d = p.seg[i], q.seg[i]
        assert s.pix_stride >= s.Cin, name   # (the packed-image stem launch has its own test: tests/test_gpu_stem_pool.py)
        x = torch.randn((s.N, s.H, s.W, s.pix_stride), generator=g, device=cuda).to(h16)
        if name.startswith("dgrad:"):
            x = x * (torch.rand((s.N, s.H, s.W, 1), generator=g, device=cuda) < 0.7).to(h16)   # gradients carry exact zeros
        K = p.R * p.S * s.Cin
        terms = s.w_terms if s.w_terms > 1 else 1
        w = torch.randn((p.R, p.S, s.Cin, s.Cout), generator=g, device=cuda) * ((0.01 if f32 else 1.0) / math.sqrt(K))
        cinp = lib.rn_conv_cin_pad(s.Cin)
        wp = torch.empty((lib.rn_conv_cout_pad(s.Cout), p.R, p.S, terms * cinp), dtype=h16, device=cuda)
        if s.w_pair:     # the two planes of an f32 kernel stacked along Cout (narrow prediction conv)
            assert terms == 1 and f32, name
            wp = torch.empty((lib.rn_conv_pair_rows(s.Cout), p.R, p.S, cinp), dtype=h16, device=cuda)
            _C.check(lib.rn_pack_conv_weight_pair(_C.ptr(w), 0, p.R, p.S, s.Cin, s.Cout, cinp, _C.ptr(wp), _C.current_stream()))
            terms = 2    # the reference below: both planes
        elif terms > 1:
            _C.check(lib.rn_pack_conv_weight_split(_C.ptr(w), 0, p.R, p.S, s.Cin, s.Cout, cinp, terms, _C.ptr(wp),
                                                   _C.current_stream()))
        else:
            _C.check(lib.rn_pack_conv_weight(_C.ptr(w), p.R, p.S, s.Cin, s.Cout, cinp, _C.ptr(wp), _C.current_stream()))
        y = torch.empty((s.N, s.Ho, s.Wo, s.Cout), dtype=torch.float32 if f32 else h16, device=cuda)
        t = dict(x=x, w=w, y=y, terms=terms)
        if s.scale:
            t["scale"] = torch.rand((s.Cout,), generator=g, device=cuda) + 0.5
        if s.shift:
            t["shift"] = torch.randn((s.Cout,), generator=g, device=cuda) * 0.1
        if s.bias:
            t["bias"] = torch.randn((s.Cout,), generator=g, device=cuda) * (1.0 if not f32 else 0.1) - (4.595 if f32 else 0.0)
        if s.residual:
            t["residual"] = torch.randn((s.N, s.Ho, s.Wo, s.Cout), generator=g, device=cuda).to(h16)
            if s.residual == s.y:            # the accumulating data-gradient launches: residual = the gradient buffer itself
                y.copy_(t["residual"])
        P = s.N * s.Ho * s.Wo
        chunks = lib.rn_conv_bn_row_blocks(ctypes.byref(p), i)     # (rows // 128) * ceil(P / rows), or two per balanced tile
        assert chunks >= (rows // 128) * ((P + rows - 1) // rows), name
        if s.bn_partial:
            t["partial"] = torch.full((chunks, 2, s.Cout), float("nan"), dtype=torch.float32, device=cuda)
        if s.bn_bwd_y:
            t["bn_y"] = (torch.randn((s.N, s.Ho, s.Wo, s.Cout), generator=g, device=cuda) * 1.5).to(h16)
            mean = torch.randn((s.Cout,), generator=g, device=cuda) * 0.3
            invstd = torch.rand((s.Cout,), generator=g, device=cuda) + 0.5
            gamma = torch.rand((s.Cout,), generator=g, device=cuda) + 0.5
            beta = torch.randn((s.Cout,), generator=g, device=cuda) * 0.3
            t["bn_fwd"] = torch.stack([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
        d.x, d.w, d.y = x.data_ptr(), wp.data_ptr(), y.data_ptr()
        d.scale = t["scale"].data_ptr() if "scale" in t else None
        d.shift = t["shift"].data_ptr() if "shift" in t else None
        d.bias = t["bias"].data_ptr() if "bias" in t else None
        d.residual = (y.data_ptr() if s.residual == s.y else t["residual"].data_ptr()) if s.residual else None
        d.bn_partial = t["partial"].data_ptr() if "partial" in t else None
        d.bn_bwd_y = t["bn_y"].data_ptr() if "bn_y" in t else None
        d.bn_bwd_fwd = t["bn_fwd"].data_ptr() if "bn_fwd" in t else None
        d.w_terms, d.w_pair = s.w_terms, s.w_pair
        d.N, d.H, d.W, d.Cin, d.pix_stride, d.Ho, d.Wo, d.Cout = s.N, s.H, s.W, s.Cin, s.pix_stride, s.Ho, s.Wo, s.Cout
        keep.append(wp)
        per_seg.append(t)
    ws = None
    if p.splitk_ws:      # the engine's launches carry a split-K workspace (the last round of a persistent grid is split along K)
        ws = torch.zeros((int(p.splitk_ws_bytes),), dtype=torch.uint8, device=cuda)
        q.splitk_ws, q.splitk_ws_bytes = ws.data_ptr(), ws.numel()
    assert lib.rn_conv_kernel_id(ctypes.byref(q)) == lib.rn_conv_kernel_id(ctypes.byref(p)), name
    assert lib.rn_conv_splitk_workspace_bytes(ctypes.byref(q)) == lib.rn_conv_splitk_workspace_bytes(ctypes.byref(p)), name
    _C.check(lib.rn_conv2d_nhwc_fwd(ctypes.byref(q), _C.current_stream()), name)
    torch.cuda.synchronize()
    if ws is not None:   # every arrival counter consumed by the last arriver
        assert int(ws[:16384].view(torch.int32).abs().sum().item()) == 0, name
    for i, t in enumerate(per_seg):
        s = p.seg[i]
        xp = _pad_input(t["x"][..., :s.Cin], p.R, p.S, p.stride_h, p.pad_top, p.pad_left, s.Ho, s.Wo)
        # the weights the kernel multiplies by: the 16-bit rounding of w (one plane), or the sum of its split planes
        w16 = t["w"].to(h16).double()
        if t["terms"] >= 2:
            r1 = t["w"] - t["w"].to(h16).float()
            w16 = w16 + r1.to(h16).double()
            if t["terms"] == 3:
                w16 = w16 + (r1 - r1.to(h16).float()).to(h16).double()
        acc = torch.zeros((s.N * s.Ho * s.Wo, s.Cout), dtype=torch.float64, device=cuda)
        for r, c, v in _tap_views(xp, p.R, p.S, p.stride_h, s.Ho, s.Wo):
            acc += v.reshape(-1, s.Cin).double() @ w16[r, c]
        yv = acc.reshape(s.N, s.Ho, s.Wo, s.Cout)
        del acc
        if "bias" in t:
            yv = yv + t["bias"].double()
        affine = "scale" in t or "shift" in t
        rb = (lambda v: v) if f32 else (lambda v: _rb(v, h16))
        if affine or "residual" in t:
            yv = rb(yv)
        if "scale" in t:
            yv = yv * t["scale"].double()
        if "shift" in t:
            yv = yv + t["shift"].double()
        if "residual" in t:
            yv = (rb(yv) if affine else yv) + t["residual"].double()
        if p.act == _C.RN_ACT_RELU:
            yv = F.relu(yv)
        elif p.act == _C.RN_ACT_RELU6:
            yv = F.relu6(yv)
        elif p.act == _C.RN_ACT_SWISH:
            yv = rb(yv)
            yv = yv * torch.sigmoid(yv)
        got = t["y"].float()
        if f32:
            want = yv.float()
            spread = (want - want.mean()).abs().max().item() + 1e-6
            assert (got - want).abs().max().item() <= 2e-4 * spread, (name, i)
        else:
            _close_bf16(got, rb(yv).float())
        stored = t["y"].double().reshape(-1, s.Cout)
        if "partial" in t and "bn_y" not in t:      # forward statistics of the STORED tensor
            part = t["partial"].double()
            assert torch.isfinite(part).all(), (name, i)
            sums = part.sum(0)
            want = torch.stack([stored.sum(0), (stored * stored).sum(0)])
            torch.testing.assert_close(sums, want, rtol=2e-5, atol=2e-5 * want.abs().max().item())
        if "bn_y" in t:                              # stage 1 of the BatchNorm backward reduction on the stored dz
            part = t["partial"].double()
            assert torch.isfinite(part).all(), (name, i)
            mean, invstd, sc, sh = t["bn_fwd"]
            yb = t["bn_y"].float().reshape(-1, s.Cout)
            gate = ((yb * sc + sh) > 0).double()     # the kernels' fp32 mask
            gg = stored * gate
            want = torch.stack([gg.sum(0), (gg * (yb.double() - mean.double()) * invstd.double()).sum(0)])
            torch.testing.assert_close(part.sum(0), want, rtol=1e-4, atol=2e-5 * want.abs().max().item())
        del yv, got, stored


def _check_wgrad_launch(cuda, eng, name, p):
    from retinanet import _C
    lib, h16 = eng.lib, eng.h16
    g = torch.Generator(device=cuda).manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
    q = _C.WgradProblem()
    q.R, q.S, q.stride_h, q.stride_w, q.pad_top, q.pad_left = p.R, p.S, p.stride_h, p.stride_w, p.pad_top, p.pad_left
    q.num_segments, q.opts = p.num_segments, p.opts
    cin, cout = p.seg[0].Cin, p.seg[0].Cout
    want = torch.zeros((cout, p.R, p.S, cin), dtype=torch.float64, device=cuda)
    keep = []
    for i in range(p.num_segments):
        s, d = p.seg[i], q.seg[i]
        xs = s.x_pix_stride if s.x_pix_stride > 0 else s.Cin
        dys = s.dy_pix_stride if s.dy_pix_stride > 0 else s.Cout
        assert xs >= s.Cin, name
        x = torch.randn((s.N, s.H, s.W, xs), generator=g, device=cuda).relu().to(h16)        # saved activations are post-ReLU
        dy = (torch.randn((s.N, s.Ho, s.Wo, dys), generator=g, device=cuda)
              * (torch.rand((s.N, s.Ho, s.Wo, 1), generator=g, device=cuda) < 0.7)).to(h16)
        d.x, d.dy = x.data_ptr(), dy.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = s.N, s.H, s.W, s.Cin, s.Ho, s.Wo, s.Cout
        d.dy_pix_stride, d.x_pix_stride = s.dy_pix_stride, s.x_pix_stride
        keep += [x, dy]
        xp = _pad_input(x[..., :s.Cin], p.R, p.S, p.stride_h, p.pad_top, p.pad_left, s.Ho, s.Wo)
        dym = dy[..., :s.Cout].reshape(-1, s.Cout).double().t().contiguous()
        for r, c, v in _tap_views(xp, p.R, p.S, p.stride_h, s.Ho, s.Wo):
            want[:, r, c, :] += dym @ v.reshape(-1, s.Cin).double()
        del dym, xp
    nbytes = lib.rn_wgrad_workspace_bytes(ctypes.byref(q))
    assert nbytes == lib.rn_wgrad_workspace_bytes(ctypes.byref(p)), name     # the same split-K plan as the engine's launch
    ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=cuda)
    ws.fill_(0x7f)
    dw = torch.full((cout, p.R, p.S, cin), 7.0, dtype=torch.float32, device=cuda)
    _C.check(lib.rn_conv2d_nhwc_wgrad(ctypes.byref(q), _C.ptr(dw), 0.0, _C.ptr(ws), ws.numel(), _C.current_stream()), name)
    torch.cuda.synchronize()
    first = dw.clone()
    scale = want.abs().max().item()
    torch.testing.assert_close(dw.double(), want, rtol=1e-3, atol=1e-3 * scale)
    # deterministic: a second launch gives the same bits (ordered split-K reduction)
    _C.check(lib.rn_conv2d_nhwc_wgrad(ctypes.byref(q), _C.ptr(dw), 0.0, _C.ptr(ws), ws.numel(), _C.current_stream()), name)
    torch.cuda.synchronize()
    assert torch.equal(dw, first), name


@pytest.mark.parametrize("size,B,splitk_env", [(640, 32, "whole"), (1024, 32, "whole"), (640, 32, "split")],
                         ids=["bench-640-b32", "bench-1024-b32", "bench-640-b32-splitk"], indirect=["splitk_env"])
def test_every_distinct_launch_of_the_bench_engine(cuda, size, B, splitk_env):
    eng = _engine(cuda, size, B)
    convs, seen = [], set()
    for name, p in eng.conv_launches:
        sig = _conv_sig(eng, name, p)
        if sig not in seen:
            seen.add(sig)
            convs.append((name, p))
    wgrads, seen = [], set()
    for name, p in eng.wgrad_launches:
        sig = _wgrad_sig(name, p)
        if sig not in seen:
            seen.add(sig)
            wgrads.append((name, p))
    kinds = {}
    n_split = 0
    for name, p in convs:
        k = (name.split(":")[0], eng.lib.rn_conv_kernel_id(ctypes.byref(p)))
        kinds[k] = kinds.get(k, 0) + 1
        n_split += int(bool(p.splitk_ws) and eng.lib.rn_conv_splitk_workspace_bytes(ctypes.byref(p)) > 0)
    # (round 4 sent the stage-4 3x3 launches — 100 tiles of 256 rows — to the halo kernel's split tiles when a workspace was
    # attached; since round 5 they stay on the 128-row kernel, whose 400 tiles need no split: at the bench batch the "split"
    # mode only proves that attached workspaces change nothing)
    assert n_split == 0, n_split
    # the engine at this size runs all three forward kernel families, as forward and as data-gradient launches; the halo
    # kernel in its 512 x 128 form (kernel id 3: the dispatcher's choice wherever the channel count is a multiple of 128)
    halo = 3
    for k in (("fwd", 0), ("fwd", 1), ("fwd", halo), ("dgrad", 0), ("dgrad", 1), ("dgrad", halo)):
        assert kinds.get(k, 0) >= 1, kinds
    pair = [eng.lib.rn_conv_kernel_id(ctypes.byref(p)) for _, p in convs if p.seg[0].w_pair]
    assert pair == [3], pair      # the box prediction conv: two weight planes as GEMM columns on the 512 x 128 tiles
    assert len(convs) >= 40 and len(wgrads) >= 25, (len(convs), len(wgrads))
    # free the engine's tensors before the float64 references are built
    eng_lib, eng_h16 = eng.lib, eng.h16
    problems = [(n, p) for n, p in convs], [(n, p) for n, p in wgrads]

    class _Stub:      # what the per-launch checks need of the engine
        lib, h16 = eng_lib, eng_h16
    keep_structs = eng._keep       # the descriptors stay alive; drop the big tensors
    for attr in ("t", "raw", "grad"):
        if hasattr(eng, attr):
            getattr(eng, attr).clear()
    del eng
    torch.cuda.empty_cache()
    failures = []
    for name, p in problems[0]:
        try:
            _check_conv_launch(cuda, _Stub, name, p)
        except AssertionError as e:
            failures.append((name, str(e).splitlines()[0][:200] if str(e) else "assert"))
        torch.cuda.empty_cache()
    for name, p in problems[1]:
        try:
            _check_wgrad_launch(cuda, _Stub, name, p)
        except AssertionError as e:
            failures.append((name, str(e).splitlines()[0][:200] if str(e) else "assert"))
        torch.cuda.empty_cache()
    del keep_structs
    print(f"{size}x{size} B={B}: {len(problems[0])} distinct conv launches {kinds}, {len(problems[1])} distinct weight-gradient "
          f"launches checked")
    assert not failures, failures
