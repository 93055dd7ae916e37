"""Timing of the stage-2 column reduction alone (rn_bn_stats on external partials) at the partial counts of the bench step.
RNET_BN_FINAL=1|2|3 python tools/probes/bn_final_probe.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch
from retinanet import _C

lib = _C.lib()
dev = torch.device("cuda:0")
CASES = [("head 5 levels C256", [(256, c) for c in (1600, 400, 100, 26, 8)]),
         ("heads 10 segs C256", [(256, c) for c in (1600, 400, 100, 26, 8)] * 2),
         ("C128 6400", [(128, 6400)]), ("C64 6400", [(64, 6400)]), ("C256 6400", [(256, 6400)]),
         ("C256 1600", [(256, 1600)]), ("C256 400", [(256, 400)]), ("C512 1600", [(512, 1600)]), ("C1024 400", [(1024, 400)])]
for name, segs in CASES:
    p = _C.BnProblem()
    p.num_segments, p.act, p.bessel, p.eps, p.momentum, p.count_scale = len(segs), 0, 0, 1e-3, 0.9, 1.0
    sums = []
    for i, (C, ch) in enumerate(segs):
        sums.append(torch.zeros((2, C), dtype=torch.float32, device=dev))
        q = p.seg[i]
        q.y, q.sums, q.P, q.C, q.ext_chunks = sums[-1].data_ptr(), sums[-1].data_ptr(), ch * 128, C, ch
    nbytes = lib.rn_bn_workspace_bytes(ctypes.byref(p))
    nbuf = max(2, int(400e6 // nbytes))
    npart = sum(2 * C * ch for C, ch in segs)
    bufs = []
    for _ in range(min(nbuf, 64)):   # random partial sums, zero tail (ticket counters)
        b = torch.zeros((nbytes // 4,), device=dev)
        b[:npart].normal_()
        bufs.append(b)
    st = _C.current_stream()
    res = {}
    for mode in ("hot", "cold"):
        for _ in range(3):
            _C.check(lib.rn_bn_stats(ctypes.byref(p), bufs[0].data_ptr(), nbytes, st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for k in range(n):
            b = bufs[0] if mode == "hot" else bufs[k % len(bufs)]
            _C.check(lib.rn_bn_stats(ctypes.byref(p), b.data_ptr(), nbytes, st))
        e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / n * 1e3
    print("%-22s %6.2f MB  hot %6.1f us  cold %6.1f us" % (name, nbytes / 1e6, res["hot"], res["cold"]))
