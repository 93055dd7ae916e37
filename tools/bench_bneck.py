"""Timing of rn_bottleneck64_fwd (one launch per ResNet stage-1 bottleneck block, csrc/rn_bneck.hip) at the 640 x 640
geometry: 160 x 160 pixels, batch 32 / 8 / 1, identity (Cx = 256) and projection (Cx = 64) blocks; algorithmic bytes =
block input + block output.  python tools/bench_bneck.py [--iters 30]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch  # noqa: E402

from retinanet import _C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--hw", type=int, default=160)
    ap.add_argument("--ablate", type=int, default=0, help="rn_launch_opts.ablate: timing probes of rn_bneck.hip (wrong results)")
    ap.add_argument("--batches", default="32,8,1")
    a = ap.parse_args()
    lib = _C.lib()
    dev = torch.device("cuda:0")
    for Cx in (256, 64):
        for B in [int(v) for v in a.batches.split(',')]:
            H = W = a.hw
            x = torch.relu(torch.randn((B, H, W, Cx), device=dev)).to(torch.bfloat16)
            y = torch.empty((B, H, W, 256), dtype=torch.bfloat16, device=dev)
            ws = [torch.randn(s, device=dev) * 0.05 for s in ((1, 1, Cx, 64), (3, 3, 64, 64), (1, 1, 64, 256))]
            wsc = torch.randn((1, 1, 64, 256), device=dev) * 0.05 if Cx == 64 else None
            packed = torch.empty((lib.rn_bottleneck64_packed_bytes(Cx),), dtype=torch.uint8, device=dev)
            _C.check(lib.rn_bottleneck64_pack(_C.ptr(ws[0]), _C.ptr(ws[1]), _C.ptr(ws[2]), _C.ptr(wsc), Cx, _C.ptr(packed),
                                              _C.current_stream()))
            aff = torch.cat([torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1] * 2 +
                            [torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1] * 2).contiguous()
            p = _C.Bottleneck64Problem()
            p.x, p.y, p.w_packed, p.affine = x.data_ptr(), y.data_ptr(), packed.data_ptr(), aff.data_ptr()
            p.N, p.H, p.W, p.Cx = B, H, W, Cx
            p.opts = _C.LaunchOpts(ablate=a.ablate)
            st = _C.current_stream()
            for _ in range(3):
                _C.check(lib.rn_bottleneck64_fwd(ctypes.byref(p), st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                _C.check(lib.rn_bottleneck64_fwd(ctypes.byref(p), st))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            byts = x.numel() * 2 + y.numel() * 2
            fl = 2 * B * H * W * (Cx * 64 + 9 * 64 * 64 + 64 * 256 + (64 * 256 if Cx == 64 else 0))
            print(f"Cx={Cx:3d} B={B:2d} {H}x{W}: {us:8.1f} us  {byts / us / 1e3:7.1f} GB/s  {fl / us / 1e6:7.1f} TFLOP/s "
                  f"({byts / 1e6:.0f} MB, {fl / 1e9:.1f} GFLOP)", flush=True)


if __name__ == "__main__":
    main()
