"""Within-process A/B timing of the weight-gradient kernels on the bench's shapes (run on the GPU box).

    python tools/bench_wgrad.py [--preset tower,class,g3,g2,fpn80] [--variants halo,big,halo:1] [--rounds 7] [--iters 5]

A variant is `<kernel>[:<ablate>[:<wgrad_target_blocks>]]`: kernel = halo (rn_launch_opts.wgrad_kernel 2), big (3: the per-tap 256-wide kernel),
small (1: the 128-tile kernel); ablate = rn_launch_opts.ablate bits the kernel under test understands (code variants
compiled side by side for A/B).  Variants are timed in interleaved rounds in ONE process (cdna_hip_programming.md 5.4
rules 13 / 24: the boxes of the pool differ by several percent, and so do separate invocations); median and min per
variant, TFLOP/s of the layer's algorithmic FLOPs; the split-K reduction launch is part of the call, like in the step."""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
from retinanet import _C  # noqa: E402

B = 32
PRESETS = {   # list of (N, H, W, Cin, Cout) segments, k, stride
    "tower": ([(B, s, s, 256, 256) for s in (80, 40, 20, 10, 5)], 3, 1),
    "class": ([(B, s, s, 256, 720) for s in (80, 40, 20, 10, 5)], 3, 1),
    "fpn80": ([(B, 80, 80, 256, 256)], 3, 1),
    "tower8": ([(B, s, s, 256, 2048) for s in (80, 40, 20, 10, 5)], 3, 1),   # 8 tower layers' worth of tiles: 4 chunks
    "g2x3": ([(B, 80, 80, 128, 384)], 3, 1),
    "g3": ([(B, 40, 40, 256, 256)], 3, 1),
    "g2": ([(B, 80, 80, 128, 128)], 3, 1),
    "g4": ([(B, 20, 20, 512, 512)], 3, 1),
    "g2_1x1": ([(B, 80, 80, 128, 512)], 1, 1),
    "g3_1x1": ([(B, 40, 40, 1024, 256)], 1, 1),
    "g3_1x1b": ([(B, 40, 40, 256, 1024)], 1, 1),
    "g2_1x1b": ([(B, 80, 80, 512, 128)], 1, 1),
    "g4_1x1": ([(B, 20, 20, 2048, 512)], 1, 1),
    "fpn_lat": ([(B, 40, 40, 1024, 256)], 1, 1),
}
KERNEL = {"halo": 2, "big": 3, "small": 1, "auto": 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="tower,class,g3,g2")
    ap.add_argument("--variants", default="halo,big")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (clock ceiling; not a throughput claim)")
    a = ap.parse_args()
    lib = _C.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for name in a.preset.split(","):
        segs, k, stride = PRESETS[name]
        pad = (k - 1) // 2
        keep, flops = [], 0
        problems = []
        for var in a.variants.split(","):
            kern, abl, tb = (var.split(":") + ["", ""])[:3]
            p = _C.WgradProblem()
            p.R = p.S = k
            p.stride_h = p.stride_w = stride
            p.pad_top = p.pad_left = pad
            p.num_segments = len(segs)
            p.opts = _C.LaunchOpts(wgrad_kernel=KERNEL[kern], ablate=int(abl or 0), wgrad_target_blocks=int(tb or 0))
            problems.append((var, p))
        cin, cout = segs[0][3], segs[0][4]
        for i, (N, H, W, ci, co) in enumerate(segs):
            Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
            if a.zeros:
                x = torch.zeros((N, H, W, ci), dtype=torch.bfloat16, device=dev)
                dy = torch.zeros((N, Ho, Wo, co), dtype=torch.bfloat16, device=dev)
            else:
                x = torch.randn((N, H, W, ci), generator=g, device=dev).relu().to(torch.bfloat16)
                dy = torch.randn((N, Ho, Wo, co), generator=g, device=dev).to(torch.bfloat16)
            keep += [x, dy]
            flops += 2 * N * Ho * Wo * k * k * ci * co
            for _, p in problems:
                s = p.seg[i]
                s.x, s.dy = x.data_ptr(), dy.data_ptr()
                s.N, s.H, s.W, s.Cin, s.Ho, s.Wo, s.Cout = N, H, W, ci, Ho, Wo, co
        dw = torch.zeros((cout, k, k, cin), dtype=torch.float32, device=dev)
        wss = []
        for _, p in problems:
            wss.append(torch.empty((max(lib.rn_wgrad_workspace_bytes(ctypes.byref(p)), 256),), dtype=torch.uint8, device=dev))
        st = _C.current_stream()
        times = {v: [] for v, _ in problems}
        ref = None
        for rnd in range(a.rounds + 1):
            for (var, p), ws in zip(problems, wss):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    _C.check(lib.rn_conv2d_nhwc_wgrad(ctypes.byref(p), _C.ptr(dw), 0.0, _C.ptr(ws), ws.numel(), st), var)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[var].append(e0.elapsed_time(e1) / a.iters * 1e3)
                elif ref is None:
                    ref = dw.clone()
                elif int((var.split(":") + [""])[1] or 0) < 2:   # (ablate bits >= 2 skip work: timing only)
                    # every variant computes the same sums (fp32 association differs between kernels)
                    err = (dw - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
                    assert err < 2e-3, (var, err)
        print(f"{name:8s} k{k} s{stride} {cin}->{cout} {flops / 1e9:8.1f} GFLOP")
        for var, p in problems:
            t = times[var]
            med, mn = statistics.median(t), min(t)
            kid = lib.rn_wgrad_kernel_id(ctypes.byref(p))
            print(f"    {var:10s} kernel {kid}: median {med:8.1f} us  min {mn:8.1f} us   {flops / med / 1e6:7.1f} TFLOP/s (median)")


if __name__ == "__main__":
    main()
