"""Sharp parity AT THE BENCH'S SHAPES for what tests/test_gpu_bench_shapes.py does not re-issue:

  (i)   every GROUPED weight-gradient launch of the B = 32 engine (`eng.wgrad_groups`: the eight head-tower layers, the
        3x3 layers of a ResNet stage) through rn_conv2d_nhwc_wgrad_group, under both split-K plans the engine uses
        (wgrad_target_blocks = the two-stream cap, and 0), against float64 per layer, bit-equal across repeats, and
        against the per-layer entry point (another split-K plan: same products, other summation order);
  (ii)  the BatchNorm passes — rn_bn_stats_finalize, rn_bn_apply (chunked multi-block form, with and without the relu
        bit mask), rn_bn_bwd_reduce, rn_bn_bwd_apply — at the engine's own rn_bn_problem geometries (419 MB stage-1
        tensors, the ten-segment head groups) against float64 (model/utils.py:7-22 and its autodiff);
  (iii) rn_retinanet_loss_fwd_bwd_bf16 and rn_anchor_match_encode at B = 32 against the oracle (loss_impl.py:15-105,
        label_encoder.py:27-125): matches / targets bit-exact, losses to 1e-5, bf16 gradients = the rounding of the
        float64 gradient to within one bf16 step.
The float64 references run on the GPU (torch / rocBLAS dgemm): an independent code path."""
import ctypes
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle as o
from test_gpu_bench_shapes import _engine, _pad_input, _tap_views, _wgrad_sig

pytestmark = pytest.mark.gpu


def _drop_engine_tensors(eng):
    keep = eng._keep
    for attr in ("t", "raw", "grad"):
        if hasattr(eng, attr):
            getattr(eng, attr).clear()
    return keep


# ---- (i) grouped weight gradients ---------------------------------------------------------------------------------------
def _wgrad_layer(cuda, g, p, h16):
    """fresh tensors of one layer's geometry + its float64 weight gradient"""
    from retinanet import _C
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
        x = torch.randn((s.N, s.H, s.W, xs), generator=g, device=cuda).relu().to(h16)
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
    return q, want, keep


def test_grouped_weight_gradient_launches_at_bench_shape(cuda):
    from retinanet import _C
    eng = _engine(cuda, 640, 32)
    lib, h16 = eng.lib, eng.h16
    groups = [list(g) for g in eng.wgrad_groups]
    assert sorted(len(g) for g in groups)[-1] == 8 and len(groups) >= 3, [len(g) for g in groups]
    cap = max(int(p.opts.wgrad_target_blocks) for g in groups for p in g)
    assert cap > 0        # the two-stream step caps the persistent weight-gradient grids (RNET_WGRAD_CUS)
    keep_structs = _drop_engine_tensors(eng)
    del eng
    torch.cuda.empty_cache()
    for gi, grp in enumerate(groups):
        n = len(grp)
        g = torch.Generator(device=cuda).manual_seed(zlib.crc32(repr(_wgrad_sig("", grp[0])).encode()) % (2 ** 31) + gi)
        layers = [_wgrad_layer(cuda, g, p, h16) for p in grp]
        per_layer = []
        for q, want, _ in layers:       # the per-layer entry point on the same tensors
            nb = lib.rn_wgrad_workspace_bytes(ctypes.byref(q))
            ws = torch.empty((max(nb, 256),), dtype=torch.uint8, device=cuda)
            dw = torch.empty(want.shape, dtype=torch.float32, device=cuda)
            _C.check(lib.rn_conv2d_nhwc_wgrad(ctypes.byref(q), _C.ptr(dw), 0.0, _C.ptr(ws), ws.numel(), _C.current_stream()))
            per_layer.append(dw)
        for target in (cap, 0):
            for q, _, _ in layers:
                q.opts.wgrad_target_blocks = target
            arr = (ctypes.POINTER(_C.WgradProblem) * n)(*[ctypes.pointer(q) for q, _, _ in layers])
            assert lib.rn_wgrad_group_fused(arr, n) == 1
            ws = torch.empty((max(lib.rn_wgrad_group_workspace_bytes(arr, n), 256),), dtype=torch.uint8, device=cuda)
            ws.fill_(0x7f)
            runs = []
            for rep_ in range(2):
                dws = [torch.full(want.shape, 7.0, dtype=torch.float32, device=cuda) for _, want, _ in layers]
                _C.check(lib.rn_conv2d_nhwc_wgrad_group(arr, n, _C.ptr_array(dws), 0.0, _C.ptr(ws), ws.numel(),
                                                        _C.current_stream()), f"group {gi} target {target}")
                torch.cuda.synchronize()
                runs.append(dws)
            for li, (q, want, _) in enumerate(layers):
                scale = want.abs().max().item()
                torch.testing.assert_close(runs[0][li].double(), want, rtol=1e-3, atol=1e-3 * scale)
                assert torch.equal(runs[0][li], runs[1][li]), (gi, target, li)        # ordered reduction: same bits
                # the per-layer launch sums other pixel chunks: equal to fp32 summation accuracy, not bit for bit
                torch.testing.assert_close(runs[0][li], per_layer[li], rtol=1e-4, atol=1e-5 * scale)
        del layers, per_layer, runs
        torch.cuda.empty_cache()
    del keep_structs


# ---- (ii) BatchNorm passes ----------------------------------------------------------------------------------------------
def _bn_sig(p):
    return (p.num_segments, p.act, p.bessel,
            tuple((int(s.P), int(s.C), bool(s.residual), bool(s.act_mask), bool(s.dres), int(s.dres_accumulate),
                   bool(s.sample_scale)) for s in (p.seg[i] for i in range(p.num_segments))))


def _act_fwd_gate(v, act):
    """(activation output, gate computed from the STORED 16-bit output like the kernels do) for relu / relu6 / none"""
    from retinanet import _C
    if act == _C.RN_ACT_RELU:
        return F.relu(v)
    if act == _C.RN_ACT_RELU6:
        return F.relu6(v)
    assert act == _C.RN_ACT_NONE
    return v


def _check_bn_problem(cuda, lib, h16, p, name):
    from retinanet import _C
    g = torch.Generator(device=cuda).manual_seed(zlib.crc32(repr(_bn_sig(p)).encode()) % (2 ** 31))
    q = _C.BnProblem()
    q.num_segments, q.act, q.bessel, q.eps, q.momentum, q.count_scale = p.num_segments, p.act, p.bessel, p.eps, p.momentum, 1.0
    T = []
    for i in range(p.num_segments):
        s, d = p.seg[i], q.seg[i]
        P, C = int(s.P), int(s.C)
        t = {"y": (torch.randn((P, C), generator=g, device=cuda) * 1.5 + 0.3).to(h16),
             "dz": (torch.randn((P, C), generator=g, device=cuda)
                    * (torch.rand((P, 1), generator=g, device=cuda) < 0.8)).to(h16),
             "gamma": torch.rand((C,), generator=g, device=cuda) + 0.5, "beta": torch.randn((C,), generator=g, device=cuda) * 0.3,
             "mm": torch.randn((C,), generator=g, device=cuda), "mv": torch.rand((C,), generator=g, device=cuda) + 0.5}
        t["mm0"], t["mv0"] = t["mm"].clone(), t["mv"].clone()
        t["z"], t["dy"] = torch.empty_like(t["y"]), torch.empty_like(t["y"])
        if s.residual:
            t["res"] = torch.randn((P, C), generator=g, device=cuda).to(h16)
        if s.dres:
            t["dres"] = torch.randn((P, C), generator=g, device=cuda).to(h16)
            t["dres0"] = t["dres"].clone()
        if s.act_mask:
            t["mask"] = torch.full((P * C // 8,), 0xA5, dtype=torch.uint8, device=cuda)
        for k, shape in (("sums", (2, C)), ("bsums", (2, C)), ("fwd", (4, C))):
            t[k] = torch.zeros(shape, dtype=torch.float32, device=cuda)
        t["dgamma"], t["dbeta"] = torch.zeros((C,), device=cuda), torch.zeros((C,), device=cuda)
        d.y, d.z, d.dz, d.dy = t["y"].data_ptr(), t["z"].data_ptr(), t["dz"].data_ptr(), t["dy"].data_ptr()
        d.residual = t["res"].data_ptr() if "res" in t else None
        d.dres = t["dres"].data_ptr() if "dres" in t else None
        d.act_mask = t["mask"].data_ptr() if "mask" in t else None
        d.sums, d.bsums, d.fwd = t["sums"].data_ptr(), t["bsums"].data_ptr(), t["fwd"].data_ptr()
        d.gamma, d.beta, d.moving_mean, d.moving_var = (t["gamma"].data_ptr(), t["beta"].data_ptr(), t["mm"].data_ptr(),
                                                        t["mv"].data_ptr())
        d.dgamma, d.dbeta = t["dgamma"].data_ptr(), t["dbeta"].data_ptr()
        d.P, d.C, d.dres_accumulate = P, C, s.dres_accumulate
        T.append(t)
    ws = torch.zeros((max(lib.rn_bn_workspace_bytes(ctypes.byref(q)), 256),), dtype=torch.uint8, device=cuda)
    st = _C.current_stream()
    _C.check(lib.rn_bn_stats_finalize(ctypes.byref(q), _C.ptr(ws), ws.numel(), st), name)
    _C.check(lib.rn_bn_apply(ctypes.byref(q), st), name)
    if all("mask" in t for t in T):
        torch.cuda.synchronize()
        zs = [t["z"].clone() for t in T]
        for t in T:
            t["z"].fill_(float("nan"))       # with the bit mask the backward passes must not look at z
    else:
        zs = [t["z"] for t in T]
    _C.check(lib.rn_bn_bwd_reduce(ctypes.byref(q), _C.ptr(ws), ws.numel(), st), name)
    _C.check(lib.rn_bn_bwd_apply(ctypes.byref(q), st), name)
    torch.cuda.synchronize()
    eps, mom = float(p.eps), float(p.momentum)
    rb = lambda v: v.float().to(h16).double()      # noqa: E731
    for i, (t, z) in enumerate(zip(T, zs)):
        y = t["y"].double()
        n = y.shape[0]
        mean = y.mean(0)
        var = (y * y).mean(0) - mean * mean
        invstd = 1.0 / torch.sqrt(var + eps)
        scale = t["gamma"].double() * invstd
        shift = t["beta"].double() - mean * scale
        fwd = t["fwd"].double()
        torch.testing.assert_close(fwd[0], mean, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(fwd[1], invstd, rtol=2e-5, atol=0)
        torch.testing.assert_close(fwd[2], scale, rtol=2e-5, atol=0)
        torch.testing.assert_close(fwd[3], shift, rtol=1e-4, atol=2e-5)
        corr = n / (n - 1.0) if p.bessel else 1.0
        torch.testing.assert_close(t["mm"].double(), t["mm0"].double() * mom + mean * (1 - mom), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(t["mv"].double(), t["mv0"].double() * mom + var * corr * (1 - mom), rtol=1e-5, atol=1e-5)
        # forward with the KERNEL'S OWN fp32 scale / shift (their accuracy is asserted above): rounding points of rn_bn_apply
        v = rb(y * fwd[2] + fwd[3])
        if "res" in t:
            v = rb(v + t["res"].double())
        want_z = _act_fwd_gate(v, p.act)
        got_z = z.double()
        tol = want_z.abs().max().item() / 128       # one 16-bit step in the top binade
        assert (got_z - want_z).abs().max().item() <= tol, (name, i, "z", (got_z - want_z).abs().max().item(), tol)
        # an fp32 multiply-add against float64 at a rounding boundary
        assert (got_z != want_z).double().mean().item() < 0.02, (name, i, "z mismatches", (got_z != want_z).double().mean().item())
        # backward from the STORED z (what the kernels gate on)
        if p.act == _C.RN_ACT_RELU:
            gate = (got_z > 0).double()
        elif p.act == _C.RN_ACT_RELU6:
            gate = ((got_z > 0) & (got_z < 6)).double()
        else:
            gate = torch.ones_like(got_z)
        if "mask" in t:
            bits = ((t["mask"].view(-1, 1) >> torch.arange(8, device=cuda, dtype=torch.uint8)) & 1).reshape(got_z.shape)
            assert torch.equal(bits.double(), gate), (name, i)
        gg = t["dz"].double() * gate
        xhat = (y - fwd[0]) * fwd[1]
        sg, sgx = gg.sum(0), (gg * xhat).sum(0)
        bs = t["bsums"].double()
        torch.testing.assert_close(bs[0], sg, rtol=1e-4, atol=2e-5 * sg.abs().max().item() + 1e-6)
        torch.testing.assert_close(bs[1], sgx, rtol=1e-4, atol=2e-5 * sgx.abs().max().item() + 1e-6)
        torch.testing.assert_close(t["dbeta"].double(), sg, rtol=1e-4, atol=2e-5 * sg.abs().max().item() + 1e-6)
        torch.testing.assert_close(t["dgamma"].double(), sgx, rtol=1e-4, atol=2e-5 * sgx.abs().max().item() + 1e-6)
        want_dy = fwd[2] * (gg - bs[0] / n - xhat * bs[1] / n)
        got_dy = t["dy"].double()
        scale_dy = want_dy.abs().max().item()
        assert (got_dy - rb(want_dy)).abs().max().item() <= scale_dy / 128, (name, i, "dy", (got_dy - rb(want_dy)).abs().max().item(), scale_dy)
        assert (got_dy != rb(want_dy)).double().mean().item() < 0.03, (name, i, "dy mismatches", (got_dy != rb(want_dy)).double().mean().item())
        if "dres" in t:
            want_r = rb(gg + (t["dres0"].double() if p.seg[i].dres_accumulate else 0.0))
            got_r = t["dres"].double()
            assert (got_r - want_r).abs().max().item() <= want_r.abs().max().item() / 128, (name, i, "dres")
            assert (got_r != want_r).double().mean().item() < 0.01, (name, i, "dres mismatches", (got_r != want_r).double().mean().item())
        del y, v, want_z, got_z, gg, xhat, want_dy, got_dy


def test_batchnorm_passes_at_bench_geometries(cuda):
    eng = _engine(cuda, 640, 32)
    lib, h16 = eng.lib, eng.h16
    seen, probs = set(), []
    for out, grp in eng.bn_groups.items():
        pb = grp[0]
        sig = _bn_sig(pb)
        if sig not in seen and not any(s[6] for s in sig[3]):
            seen.add(sig)
            probs.append((out, pb))
    shapes = {(s[0], s[1]) for _, pb in probs for s in _bn_sig(pb)[3]}
    print("BatchNorm geometries (P, C):", sorted(shapes))
    assert max(P * C for P, C in shapes) * 2 >= 200e6, shapes    # the 200 - 419 MB tensors of ResNet stages 1 / 2
    assert any(pb.num_segments == 10 for _, pb in probs)         # the ten-segment head groups
    assert any(_bn_sig(pb)[3][0][3] for _, pb in probs)          # a relu gate kept as a bit mask
    keep_structs = _drop_engine_tensors(eng)
    del eng
    torch.cuda.empty_cache()
    failures = []
    for name, pb in probs:
        try:
            _check_bn_problem(cuda, lib, h16, pb, name)
        except AssertionError as e:
            failures.append((name, _bn_sig(pb)[:3], str(e).splitlines()[0][:300] if str(e) else "assert"))
        torch.cuda.empty_cache()
    print(f"{len(probs)} distinct BatchNorm problems checked")
    assert not failures, failures
    del keep_structs


# ---- (iii) loss with bf16 gradient outputs, target encoding, at B = 32 ---------------------------------------------------
def test_loss_bf16_and_match_encode_at_batch_32(cuda):
    import bench
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.losses import RetinaNetLoss
    B, size, K, na = 32, 640, 80, 9
    p = default_params(input_size=size, batch_train=B)
    enc = LabelEncoder(p, device=cuda)
    gb, gc, cnt = bench.synth_ground_truth(B, size, 1337)
    t = enc.encode_batch(gb.to(cuda), gc.to(cuda), cnt.to(cuda))
    torch.cuda.synchronize()
    an = enc.anchors.boxes.cpu().numpy()
    m_gpu = t["_flat"]["matches"].cpu().numpy()
    ct_gpu = t["_flat"]["class-targets"].cpu().numpy()
    bt_gpu = t["_flat"]["box-targets"].cpu().numpy()
    npos = 0.0
    for i in range(B):
        G = int(cnt[i])
        m, ct, bt, n_i = o.encode_sample(an, gb[i, :G].numpy(), gc[i, :G].numpy())
        np.testing.assert_array_equal(m_gpu[i], m)
        np.testing.assert_array_equal(ct_gpu[i], ct)
        np.testing.assert_array_equal(bt_gpu[i].view(np.uint32), bt.view(np.uint32))
        assert t["num-positives"][i].item() == n_i
        npos += n_i
    # the loss on those targets, gradients written as bf16 into channel-padded dy tensors (768 / 64 wide)
    sizes = [80, 40, 20, 10, 5]
    g = torch.Generator(device=cuda).manual_seed(11)
    preds = {"class-predictions": {}, "box-predictions": {}}
    for lv, s in zip("34567", sizes):
        preds["class-predictions"][lv] = torch.randn((B, s, s, na * K), generator=g, device=cuda) - 4.595
        preds["box-predictions"][lv] = torch.randn((B, s, s, na * 4), generator=g, device=cuda) * 0.3
    bufs = {"class-predictions": {lv: torch.full((B, s, s, 768), 7.0, dtype=torch.bfloat16, device=cuda)
                                  for lv, s in zip("34567", sizes)},
            "box-predictions": {lv: torch.full((B, s, s, 64), 7.0, dtype=torch.bfloat16, device=cuda)
                                for lv, s in zip("34567", sizes)}}
    loss = RetinaNetLoss(K, p.loss)
    out = loss(t, preds, grads_bf16=bufs, grad_scale=1.0)
    torch.cuda.synchronize()
    normalizer = npos + 1.0
    assert out["num-anchors-matched"].item() == pytest.approx(normalizer)
    tot = {"class": 0.0, "box": 0.0}
    worst = 0.0
    for i in range(B):       # the oracle image by image (sums over images are what the loss is), same normaliser
        logits = np.concatenate([preds["class-predictions"][lv][i].reshape(-1, K).cpu().numpy() for lv in "34567"])[None]
        boxes = np.concatenate([preds["box-predictions"][lv][i].reshape(-1, 4).cpu().numpy() for lv in "34567"])[None]
        ref, dlog, dbox = o.retinanet_loss(logits, boxes, ct_gpu[i:i + 1], bt_gpu[i:i + 1], normalizer, K,
                                           alpha=float(p.loss.focal_loss.alpha), gamma=float(p.loss.focal_loss.gamma),
                                           label_smoothing=float(p.loss.focal_loss.label_smoothing),
                                           delta=float(p.loss.smooth_l1_loss.delta),
                                           box_loss_weight=float(p.loss.box_loss_weight),
                                           class_loss_weight=float(p.loss.class_loss_weight))
        tot["class"] += ref["class-loss"]
        tot["box"] += ref["box-loss"]
        got_l = np.concatenate([bufs["class-predictions"][lv][i, ..., :na * K].float().reshape(-1, K).cpu().numpy()
                                for lv in "34567"])
        got_b = np.concatenate([bufs["box-predictions"][lv][i, ..., :na * 4].float().reshape(-1, 4).cpu().numpy()
                                for lv in "34567"])
        # bf16 of a value within 1e-5 of the float64 gradient: at most one bf16 step (2^-8 relative) away from it
        for got, want in ((got_l, dlog[0]), (got_b, dbox[0])):
            err = np.abs(got - want)
            assert (err <= np.abs(want) * 2.0 ** -8 + 1e-12).all()
            worst = max(worst, float((err / (np.abs(want) + 1e-30)).max()))
    np.testing.assert_allclose(out["class-loss"].item(), tot["class"], rtol=1e-5)
    np.testing.assert_allclose(out["box-loss"].item(), tot["box"], rtol=1e-5)
    for key, live in (("class-predictions", na * K), ("box-predictions", na * 4)):
        for lv in "34567":
            assert bool((bufs[key][lv][..., live:] == 7.0).all())        # pad channels left alone
    assert worst > 0.0
