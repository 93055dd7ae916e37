"""Generate the committed golden fixtures (inputs + expected outputs) with the CPU oracle.

    python tests/golden/make_golden.py     # rewrites tests/golden/*.npz

The reference cannot run here (TensorFlow is absent), so these vectors come from
oracle/oracle.py — "parity unpinned" with respect to TF itself; they pin the oracle against
silent edits and give the GPU tests fixed inputs that do not need the oracle at all.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as o  # noqa: E402

AREAS = [1024.0, 4096.0, 16384.0, 65536.0, 262144.0]
RATIOS = [0.5, 1.0, 2.0]
SCALES = [1, 1.2599210498948732, 1.5874010519681994]


def synth_gt(rng, G, size):
    c = rng.uniform(0, size, (G, 2))
    wh = np.exp(rng.uniform(np.log(8), np.log(size * 0.8), (G, 2)))
    x1 = np.clip(c - wh / 2, 0, size)
    x2 = np.clip(c + wh / 2, 0, size)
    boxes = np.concatenate([(x1 + x2) / 2, np.maximum(x2 - x1, 1.0)], axis=1).astype(np.float32)
    cls = rng.integers(0, 80, G).astype(np.float32)
    return boxes, cls


def generate():
    out = {}
    rng = np.random.default_rng(1337)
    # anchors + match/encode on a 256x256 image (A = 12276), G in {0, 1, 7, 40}
    size = 256
    an = o.generate_anchors(size, size, 3, 7, AREAS, RATIOS, SCALES)
    d = {"anchors": an}
    for G in (0, 1, 7, 40):
        gb, gc = synth_gt(rng, G, size)
        if G == 7:
            gb[3] = gb[2]              # duplicate box -> collision on the forced anchor
            gb[5] = [900, 900, 8, 8]   # no overlap with any anchor -> grabs anchor 0
        m, ct, bt, npos = o.encode_sample(an, gb, gc)
        d.update({f"gt_boxes_{G}": gb, f"gt_cls_{G}": gc, f"matches_{G}": m, f"cls_t_{G}": ct,
                  f"box_t_{G}": bt, f"num_pos_{G}": np.float32(npos)})
    out["match_encode_256"] = d
    # post-processing on a 128x128 image, K = 6 classes, B = 2
    size, K, B = 128, 6, 2
    an = o.generate_anchors(size, size, 3, 7, AREAS, RATIOS, SCALES)
    A = an.shape[0]
    logits = rng.normal(-3.0, 1.5, (B, A, K)).astype(np.float32)
    logits[0, 5:40, 2] = logits[0, 4, 2]   # equal scores -> index tie-breaks in top-k and NMS
    enc = rng.normal(0, 0.25, (B, A, 4)).astype(np.float32)
    d = {"anchors": an, "logits": logits, "encoded": enc}
    for tag, sigma, topk in (("hard", 0.0, 300), ("soft", 0.5, 300), ("hard_notopk", 0.0, -1)):
        b, s, c, v = o.postprocess(logits, enc, an, size, size, pre_nms_top_k=topk, sigma=sigma,
                                   max_detections=20)
        d.update({f"boxes_{tag}": b, f"scores_{tag}": s, f"classes_{tag}": c, f"valid_{tag}": v})
    out["postprocess_128"] = d
    # losses: B = 2, A = 3069 (128x128), K = 6
    gb, gc = synth_gt(rng, 9, size)
    gc = np.minimum(gc, K - 1)
    m, ct, bt, npos = o.encode_sample(an, gb, gc)
    ct2 = np.stack([ct, np.roll(ct, 17)])
    ct2[1, 100:140] = -2.0
    bt2 = np.stack([bt, np.roll(bt, 17, axis=0)])
    lg = rng.normal(-4.595, 1.0, (B, A, K)).astype(np.float32)
    bp = rng.normal(0, 0.3, (B, A, 4)).astype(np.float32)
    normalizer = float(2 * npos + 1)
    losses, dl, db = o.retinanet_loss(lg, bp, ct2, bt2, normalizer, K)
    out["loss_128"] = {"logits": lg, "box_preds": bp, "cls_t": ct2, "box_t": bt2,
                       "normalizer": np.float32(normalizer),
                       "losses": np.float64([losses["box-loss"], losses["class-loss"], losses["weighted-loss"]]),
                       "dlogits": dl, "dbox": db, "anchors": an}
    return out


if __name__ == "__main__":
    for name, arrays in generate().items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        print(name, {k: getattr(v, "shape", ()) for k, v in arrays.items()})
