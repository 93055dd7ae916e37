"""CPU simulation of one soft-NMS list of the config-4 serving bench (anchors of a 640 x 640 image as boxes, class scores
sigmoid(N(-4.595, 1)) > 0.05, top 5 000, sigma 0.5 -> TF scale -2, 100 detections): counts what the GPU kernel's pop loop
does — pops, re-inserts, switches of the leading 64-candidate block, selected boxes re-checked per pop, non-unit weights.
python tools/probes/soft_nms_sim.py [--seed 0] [--lists 4]"""
import argparse, heapq, math
import numpy as np


def anchors(size=640):
    out = []
    for lvl in range(3, 8):
        stride, area = 2 ** lvl, (2 ** (lvl + 2)) ** 2
        n = size // stride
        cy, cx = np.meshgrid((np.arange(n) + 0.5) * stride, (np.arange(n) + 0.5) * stride, indexing="ij")
        for ratio in (0.5, 1.0, 2.0):
            for sc in (1.0, 2 ** (1 / 3), 2 ** (2 / 3)):
                h, w = math.sqrt(area / ratio) * sc, math.sqrt(area * ratio) * sc
                out.append(np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).reshape(-1, 4))
    return np.clip(np.concatenate(out) / size, 0, 1).astype(np.float32)


def iou(a, b):
    area_a = (a[2] - a[0]) * (a[3] - a[1]); area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    ih = np.maximum(np.minimum(a[2], b[:, 2]) - np.maximum(a[0], b[:, 0]), 0)
    iw = np.maximum(np.minimum(a[3], b[:, 3]) - np.maximum(a[1], b[:, 1]), 0)
    inter = ih * iw
    return np.where((area_a > 0) & (area_b > 0), inter / (area_a + area_b - inter), 0).astype(np.float32)


def run(seed, A, sc=None):
    if sc is None:
        rng = np.random.default_rng(seed)
        sc = (1 / (1 + np.exp(-rng.normal(-4.595, 1.0, len(A))))).astype(np.float32)
    idx = np.nonzero(sc > 0.05)[0]
    idx = idx[np.argsort(-sc[idx], kind="stable")][:5000]
    boxes, cur = A[idx], sc[idx].copy()
    n = len(idx)
    begin = np.zeros(n, np.int32)
    heap = [(-cur[i], i) for i in range(n)]
    heapq.heapify(heap)
    sel = []
    st = dict(n=n, pops=0, reinserts=0, drops=0, switches=0, fetches=0, rechecked=0, nonunit=0, lead_runs=[])
    tags = [-1, -1]
    lead, run_len = -1, 0
    while heap and len(sel) < 100:
        s, i = heapq.heappop(heap)
        s = -s
        st["pops"] += 1
        if i // 64 != lead:
            st["switches"] += 1
            lead = i // 64
            if tags[lead & 1] != lead:      # the kernel's two-slot box cache (even / odd blocks)
                tags[lead & 1] = lead
                st["fetches"] += 1
        new = np.float32(s)
        if len(sel) > begin[i]:
            js = np.arange(len(sel) - 1, begin[i] - 1, -1)
            w = np.exp(np.float32(-2.0) * iou(boxes[i], boxes[np.array(sel)[js]]) ** 2).astype(np.float32)
            st["rechecked"] += len(js)
            st["nonunit"] += int((w != 1).sum())
            for x in w:
                new = np.float32(new * x)
                if new <= 0.05:
                    break
        begin[i] = len(sel)
        if new == np.float32(s):
            sel.append(i)
        elif new > 0.05:
            st["reinserts"] += 1
            heapq.heappush(heap, (-new, i))
        else:
            st["drops"] += 1
    return st


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lists", type=int, default=4)
    ap.add_argument("--dump", default="", help="npz of tools/probes/c4_dump_lists.py: the bench's own boxes and scores")
    a = ap.parse_args()
    if a.dump:
        d = np.load(a.dump)
        for k in range(min(a.lists, d["scores"].shape[1])):
            st = run(0, d["boxes"], d["scores"][:, k].copy())
            st.pop("lead_runs")
            print("class", k, st)
    else:
        A = anchors()
        for k in range(a.lists):
            st = run(a.seed + k, A)
            st.pop("lead_runs")
            print(st)
