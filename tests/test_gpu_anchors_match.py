"""GPU parity: a1 anchors (bit-exact) and a2-a4 match/encode (matches bit-exact, targets
bit-exact thanks to rn_math.h) against the oracle and the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

import oracle as o
from make_golden import AREAS, RATIOS, SCALES, synth_gt

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gen(size, dev, params):
    from retinanet.dataloader import AnchorBoxGenerator
    return AnchorBoxGenerator(size, size, 3, 7, params.anchor_params, device=dev)


@pytest.mark.parametrize("size", [640, 1024, 896, 256, 1280])
def test_anchors_bit_exact(cuda, params, size):
    g = _gen(size, cuda, params)
    want = o.generate_anchors(size, size, 3, 7, AREAS, RATIOS, SCALES)
    got = g.boxes.cpu().numpy()
    assert g.anchor_boundaries == o.anchor_boundaries(size, size, 3, 7, 9)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_anchors_nonsquare_and_odd(cuda, params):
    from retinanet.dataloader import AnchorBoxGenerator
    g = AnchorBoxGenerator(600, 333, 3, 7, params.anchor_params, device=cuda)
    want = o.generate_anchors(600, 333, 3, 7, AREAS, RATIOS, SCALES)
    np.testing.assert_array_equal(g.boxes.cpu().numpy().view(np.uint32), want.view(np.uint32))


def _encode(params, cuda, size, gts):
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    p = default_params(input_size=size)
    enc = LabelEncoder(p, device=cuda)
    B = len(gts)
    Gmax = max(1, max(g[0].shape[0] for g in gts))
    gb = np.zeros([B, Gmax, 4], np.float32)
    gc = np.zeros([B, Gmax], np.float32)
    cnt = np.zeros([B], np.int32)
    for i, (b, c) in enumerate(gts):
        gb[i, :b.shape[0]] = b
        gc[i, :c.shape[0]] = c
        cnt[i] = b.shape[0]
    t = enc.encode_batch(torch.from_numpy(gb), torch.from_numpy(gc), torch.from_numpy(cnt))
    torch.cuda.synchronize()
    return enc, t


def test_match_encode_golden(cuda, params):
    with np.load(os.path.join(GOLD, "match_encode_256.npz")) as z:
        gts = [(z[f"gt_boxes_{G}"], z[f"gt_cls_{G}"]) for G in (0, 1, 7, 40)]
        enc, t = _encode(params, cuda, 256, gts)
        np.testing.assert_array_equal(enc.anchors.boxes.cpu().numpy(), z["anchors"])
        for i, G in enumerate((0, 1, 7, 40)):
            np.testing.assert_array_equal(t["_flat"]["matches"][i].cpu().numpy(), z[f"matches_{G}"])
            np.testing.assert_array_equal(t["_flat"]["class-targets"][i].cpu().numpy(), z[f"cls_t_{G}"])
            np.testing.assert_array_equal(t["_flat"]["box-targets"][i].cpu().numpy().view(np.uint32),
                                          z[f"box_t_{G}"].view(np.uint32))
            assert t["num-positives"][i].item() == z[f"num_pos_{G}"]


@pytest.mark.parametrize("size,Gs", [(640, [0, 1, 7, 100]), (1024, [32, 3])])
def test_match_encode_vs_oracle_full_size(cuda, params, size, Gs):
    rng = np.random.default_rng(size)
    gts = [synth_gt(rng, G, size) for G in Gs]
    if size == 640:
        gts[2][0][4] = gts[2][0][1]               # duplicate GT
        gts[2][0][6] = [5000, 5000, 10, 10]       # zero overlap -> anchor 0
    enc, t = _encode(params, cuda, size, gts)
    an = enc.anchors.boxes.cpu().numpy()
    bnd = enc.anchors.anchor_boundaries
    for i, (gb, gc) in enumerate(gts):
        m, ct, bt, npos = o.encode_sample(an, gb, gc)
        np.testing.assert_array_equal(t["_flat"]["matches"][i].cpu().numpy(), m)
        np.testing.assert_array_equal(t["_flat"]["class-targets"][i].cpu().numpy(), ct)
        np.testing.assert_array_equal(t["_flat"]["box-targets"][i].cpu().numpy().view(np.uint32), bt.view(np.uint32))
        assert t["num-positives"][i].item() == npos
        # per-level views are the reference's reshape of the boundary slices (label_encoder.py:106-116)
        for li, lv in enumerate(range(3, 8)):
            v = t["class-targets"][str(lv)][i].cpu().numpy().reshape(-1)
            np.testing.assert_array_equal(v, ct[bnd[li]:bnd[li + 1]])
            v = t["box-targets"][str(lv)][i].cpu().numpy().reshape(-1, 4)
            np.testing.assert_array_equal(v, bt[bnd[li]:bnd[li + 1]])


def test_match_properties_batch32(cuda, params):
    """BASELINE config 2 size (B=32 per GPU): properties that do not need the oracle."""
    rng = np.random.default_rng(7)
    gts = [synth_gt(rng, int(rng.integers(1, 33)), 640) for _ in range(32)]
    enc, t = _encode(params, cuda, 640, gts)
    m = t["_flat"]["matches"].cpu().numpy()
    ct = t["_flat"]["class-targets"].cpu().numpy()
    bt = t["_flat"]["box-targets"].cpu().numpy()
    for i, (gb, gc) in enumerate(gts):
        G = gb.shape[0]
        assert m[i].min() >= -1 and m[i].max() <= G - 1     # ignore band is empty at 0.5/0.5
        # force-match: GT 0 always owns its best anchor; a later GT can lose a collision (label_encoder.py:47-54)
        assert set(np.unique(m[i][m[i] >= 0])) <= set(range(G)) and (m[i] == 0).any()
        assert t["num-positives"][i].item() == (m[i] >= 0).sum()
        assert (bt[i][m[i] < 0] == 0).all()
        np.testing.assert_array_equal(ct[i][m[i] >= 0], gc[m[i][m[i] >= 0]])
