"""Micro-benchmark of rn_depthwise_conv2d_nhwc_fwd on EfficientNet-B3's depthwise layers (batch 32): forward form
(scale/shift + swish) and data-gradient form (plain).  python tools/bench_dw.py [--batch 32]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch
from retinanet import _C

SHAPES = [(3, 2, 320, 144), (3, 1, 160, 192), (5, 2, 160, 192), (3, 1, 320, 40), (3, 1, 320, 24), (5, 1, 80, 288),
          (3, 2, 80, 288), (5, 1, 40, 816), (5, 2, 40, 816), (3, 1, 80, 160), (3, 1, 40, 576), (5, 1, 40, 576),
          (3, 1, 20, 2304), (5, 1, 20, 1392)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--forms", default="fwd,dgrad,wgrad")
    a = ap.parse_args()
    lib = _C.lib()
    dev = torch.device("cuda:0")
    st = _C.current_stream()
    for k, stride, H, C in SHAPES:
        Ho = (H + stride - 1) // stride
        tot = max((Ho - 1) * stride + k - H, 0)
        x = torch.randn((a.batch, H, H, C), device=dev).to(torch.bfloat16)
        y = torch.empty((a.batch, Ho, Ho, C), dtype=torch.bfloat16, device=dev)
        w = torch.randn((k * k, C), device=dev).to(torch.bfloat16)
        sc, sh = torch.rand((C,), device=dev) + 0.5, torch.randn((C,), device=dev)
        for form in [f for f in a.forms.split(",") if f in ("fwd", "dgrad")]:
            p = _C.DwProblem()
            p.k, p.stride, p.pad_top, p.pad_left = k, stride, tot // 2, tot // 2
            p.act = _C.ACT_IDS["swish"] if form == "fwd" else 0
            p.num_segments = 1
            s = p.seg[0]
            s.x, s.w, s.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
            s.scale, s.shift = (sc.data_ptr(), sh.data_ptr()) if form == "fwd" else (None, None)
            s.residual = None
            s.N, s.H, s.W, s.C, s.Ho, s.Wo = a.batch, H, H, C, Ho, Ho
            for _ in range(3):
                _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(ctypes.byref(p), st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(ctypes.byref(p), st))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            byts = (x.numel() + y.numel()) * 2
            print(f"k{k} s{stride} {H:3d}x{H:<3d} C={C:4d} {form:5s} {us:8.1f} us  {byts / us / 1e3:7.0f} GB/s  ({byts / 1e6:.0f} MB)", flush=True)
        if "wgrad" in a.forms.split(","):     # weight gradient: x and dy read once, dW [k*k][C] f32 out
            p = _C.DwProblem()
            p.k, p.stride, p.pad_top, p.pad_left = k, stride, tot // 2, tot // 2
            p.act, p.num_segments = 0, 1
            s = p.seg[0]
            dy = torch.randn((a.batch, Ho, Ho, C), device=dev).to(torch.bfloat16)
            s.x, s.w, s.y = x.data_ptr(), None, dy.data_ptr()
            s.scale, s.shift, s.residual = None, None, None
            s.N, s.H, s.W, s.C, s.Ho, s.Wo = a.batch, H, H, C, Ho, Ho
            dw = torch.empty((k * k, C), dtype=torch.float32, device=dev)
            ws = torch.empty((max(lib.rn_depthwise_wgrad_workspace_bytes(ctypes.byref(p)), 256),), dtype=torch.uint8, device=dev)
            run = lambda: _C.check(lib.rn_depthwise_conv2d_nhwc_wgrad(ctypes.byref(p), dw.data_ptr(), ws.data_ptr(), ws.numel(), st))
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            byts = (x.numel() + dy.numel()) * 2
            print(f"k{k} s{stride} {H:3d}x{H:<3d} C={C:4d} wgrad {us:8.1f} us  {byts / us / 1e3:7.0f} GB/s  ({byts / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
