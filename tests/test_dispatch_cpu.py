"""Host logic of the convolution dispatcher (no GPU: rn_conv_kernel_id / rn_conv_tile_rows / rn_conv_bn_row_blocks /
rn_conv_splitk_workspace_bytes are pure functions of the problem descriptor; without a device the library assumes the
MI355X's 256 compute units).  Pins which kernel family, tile shape and split plan the layers of the bench configurations
get — the policy DESIGN.md section 4 describes — so that a dispatcher edit shows up as a diff here, not as a silent
slow-down on the GPU box."""
import ctypes

import pytest

from retinanet import _C

WS = 64 << 20   # a split-K workspace "is attached" (address never dereferenced by the queries)


def _problem(B, H, W, Cin, Cout, k, stride=1, ws=False, f32=False, w_terms=0, w_pair=0, segs=None):
    p = _C.ConvProblem()
    p.R = p.S = k
    p.stride_h = p.stride_w = stride
    p.pad_top = p.pad_left = (k - 1) // 2
    p.act, p.out_dtype = 0, (_C.RN_DT_F32 if f32 else _C.RN_DT_BF16)
    segs = segs or [(H, W, Cin, Cout)]
    p.num_segments = len(segs)
    for i, (h, w, ci, co) in enumerate(segs):
        s = p.seg[i]
        ho, wo = (h + 2 * p.pad_top - k) // stride + 1, (w + 2 * p.pad_left - k) // stride + 1
        s.N, s.H, s.W, s.Cin, s.pix_stride, s.Ho, s.Wo, s.Cout = B, h, w, ci, ci, ho, wo, co
        s.w_terms, s.w_pair = w_terms, w_pair
    if ws:
        p.splitk_ws, p.splitk_ws_bytes = 16, WS
    return p


def _q(p):
    lib = _C.lib()
    r = ctypes.byref(p)
    return (lib.rn_conv_kernel_id(r), lib.rn_conv_tile_rows(r), lib.rn_conv_bn_row_blocks(r, 0),
            int(lib.rn_conv_splitk_workspace_bytes(r)))


PYR = [(80, 80), (40, 40), (20, 20), (10, 10), (5, 5)]

CASES = [
    # name, problem kwargs, (kernel id, tile rows), balanced?, splits?
    ("B32 head tower 3x3 (both heads, five levels)", dict(B=32, H=0, W=0, Cin=0, Cout=0, k=3, segs=[(h, w, 256, 256) for h, w in PYR] * 2), (3, 512), False, False),
    ("B32 stage-3 3x3 256", dict(B=32, H=40, W=40, Cin=256, Cout=256, k=3), (3, 512), False, False),
    ("B32 stage-2 3x3 128", dict(B=32, H=80, W=80, Cin=128, Cout=128, k=3), (3, 512), False, False),
    ("B32 stage-1 3x3 64 (plain form)", dict(B=32, H=160, W=160, Cin=64, Cout=64, k=3), (0, 128), False, False),
    ("B32 stage-1 3x3 64 as pixel pairs", dict(B=32, H=160, W=80, Cin=128, Cout=128, k=3), (3, 512), False, False),
    ("B32 stage-2 *_out 1x1 128->512", dict(B=32, H=80, W=80, Cin=128, Cout=512, k=1), (1, 256), True, False),
    ("B32 stage-3 *_out 1x1 256->1024", dict(B=32, H=40, W=40, Cin=256, Cout=1024, k=1), (1, 256), True, False),
    ("B32 stage-3 *_a 1x1 1024->256 (K > 512: whole tiles)", dict(B=32, H=40, W=40, Cin=1024, Cout=256, k=1), (1, 256), False, False),
    ("B32 stage-1 *_out 1x1 64->256 (12.5 rounds: nothing to balance)", dict(B=32, H=160, W=160, Cin=64, Cout=256, k=1), (1, 256), False, False),
    ("B32 stage-2 *_a 1x1 512->128 (narrow: 128-row kernel)", dict(B=32, H=80, W=80, Cin=512, Cout=128, k=1), (0, 128), False, False),
    ("B32 stage-4 first 1x1 2048->512 (100 256-row tiles: 128-row kernel)", dict(B=32, H=20, W=20, Cin=2048, Cout=512, k=1), (0, 128), False, False),
    ("B32 class prediction 3x3, one plane (training)", dict(B=32, H=0, W=0, Cin=0, Cout=0, k=3, f32=True, segs=[(h, w, 256, 720) for h, w in PYR]), (3, 512), False, False),
    ("B32 box prediction 3x3, planes along Cout", dict(B=32, H=0, W=0, Cin=0, Cout=0, k=3, f32=True, w_pair=1, segs=[(h, w, 256, 36) for h, w in PYR]), (3, 512), False, False),
    ("B8 stage-4 3x3 512 with a workspace: 128-row kernel (200 tiles of 128 x 64: nothing to split)", dict(B=8, H=20, W=20, Cin=512, Cout=512, k=3, ws=True), (0, 128), False, False),
    ("B8 stage-3 3x3 256 with a workspace: 128-row kernel", dict(B=8, H=40, W=40, Cin=256, Cout=256, k=3, ws=True), (0, 128), False, False),
    ("B8 stage-4 3x3 512 without a workspace: 128-row kernel", dict(B=8, H=20, W=20, Cin=512, Cout=512, k=3), (0, 128), False, False),
    ("B1 stage-4 3x3 512: 128-row kernel, split along K", dict(B=1, H=20, W=20, Cin=512, Cout=512, k=3, ws=True), (0, 128), False, True),
    ("B1 stage-4 first 1x1 2048->512: split", dict(B=1, H=20, W=20, Cin=2048, Cout=512, k=1, ws=True), (0, 128), False, True),
    ("B1 stage-1 1x1 64->64 (one K step: nothing to split)", dict(B=1, H=160, W=160, Cin=64, Cout=64, k=1, ws=True), (0, 128), False, False),
    ("B1 head tower 3x3: 128-row kernel (the pyramid launches stay off the halo kernel's split tiles)", dict(B=1, H=0, W=0, Cin=0, Cout=0, k=3, ws=True, segs=[(h, w, 256, 256) for h, w in PYR] * 2), (0, 128), False, False),
    ("B1 FPN output 3x3: 128-row kernel", dict(B=1, H=0, W=0, Cin=0, Cout=0, k=3, ws=True, segs=[(h, w, 256, 256) for h, w in PYR]), (0, 128), False, False),
]


@pytest.mark.parametrize("name,kw,kernel,balanced,splits", CASES, ids=[c[0] for c in CASES])
def test_dispatch_of_the_bench_layers(name, kw, kernel, balanced, splits):
    p = _problem(**kw)
    kid, rows, blocks, ws = _q(p)
    assert (kid, rows) == kernel, (kid, rows)
    s = p.seg[0]
    M = s.N * s.Ho * s.Wo
    whole = (rows // 128) * -(-M // rows)
    if balanced:
        assert blocks > whole and blocks % 2 == 0, (blocks, whole)       # two (short) blocks per balanced tile
        r = 2 * M / blocks                                               # rows per tile, within rounding
        n_tiles = -(-s.Cout // 256)
        tiles = (blocks // 2) * n_tiles
        assert tiles <= -(-(-(-M // 256) * n_tiles) // 256) * 256        # the same number of rounds as whole tiles
        assert 64 <= r <= 240
    else:
        assert blocks == whole, (blocks, whole)
    assert (ws > 0) == splits, ws
    if splits:
        assert ws <= _C.lib().rn_conv_splitk_workspace_max_bytes()


def test_forced_tile_shapes_keep_whole_tiles():
    """rn_launch_opts (tests, A/B timing) switch the balancing off: the partial-sum layout of a forced launch is the plain one"""
    p = _problem(B=32, H=40, W=40, Cin=256, Cout=1024, k=1)
    assert _q(p)[2] == 512
    p.opts = _C.LaunchOpts(conv_tile=2)
    assert _q(p)[:3] == (1, 256, 400)
    p.opts = _C.LaunchOpts(max_workgroups=64)
    assert _q(p)[2] == 400


def _bn_problem(segs):
    p = _C.BnProblem()
    p.num_segments, p.act, p.bessel, p.eps, p.momentum, p.count_scale = len(segs), 0, 0, 1e-3, 0.9, 1.0
    for i, (P, C, ext, ext_bwd) in enumerate(segs):
        q = p.seg[i]
        q.P, q.C, q.ext_chunks, q.ext_chunks_bwd = P, C, ext, ext_bwd
    return p


def test_bn_workspace_layout():
    """rn_bn_workspace_bytes = [partial sums of the mode that needs more][one ticket counter per (segment, 16 channels)]
    [part slots of the split segments] (rnet_hip.h).  The partial offsets of both modes stay inside the partial region —
    the counters must never be written by a partial sum — and only segments with more than 512 rows of partials get slots."""
    lib = _C.lib()
    # a 160 x 160 stage-1 layer at batch 32: 6 400 rows of epilogue partials forward, the library's own chunking backward
    segs = [(819200, 256, 6400, 0)]
    p = _bn_problem(segs)
    need = lib.rn_bn_workspace_bytes(ctypes.byref(p))
    partial = 6400 * 2 * 256 * 4
    counters = 256 // 16 * 4
    slots = 256 // 16 * 25 * 256                       # 25 parts of 256 rows, 2 x 16 doubles each
    assert need == partial + 256 + slots, (need, partial, counters, slots)   # counters padded to 256 bytes
    # the five head levels: only the finest (1 600 rows of partials) is split
    head = [(32 * h * h, 256, 2 * -(-(32 * h * h) // 256), 0) for h in (80, 40, 20, 10, 5)]
    p = _bn_problem(head)
    need = lib.rn_bn_workspace_bytes(ctypes.byref(p))
    partial = sum(c * 2 * 256 * 4 for (_, _, c, _) in head)
    assert [c for (_, _, c, _) in head] == [1600, 400, 100, 26, 8]
    slots = 256 // 16 * 7 * 256                        # level 0: ceil(1600 / 256) = 7 parts
    assert need == (partial + 255) // 256 * 256 + 5 * 64 // 256 * 256 + 256 * (5 * 64 % 256 > 0) + slots
    for i in range(5):
        assert lib.rn_bn_partial_offset_bytes(ctypes.byref(p), i) < partial
        assert lib.rn_bn_bwd_partial_offset_bytes(ctypes.byref(p), i) < partial
    # a small layer: no slots, just the counters behind the partials of the mode that needs more of them (backward: the
    # library's own chunking, 200 chunks of 64 rows, against the 100 rows of epilogue partials forward)
    p = _bn_problem([(12800, 2048, 100, 0)])
    assert lib.rn_bn_workspace_bytes(ctypes.byref(p)) == 200 * 2 * 2048 * 4 + 2048 // 16 * 4
