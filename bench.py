"""bench.py — images/s of the RetinaNet hot path on MI355X (BASELINE.json metric:
"images/sec train+infer, ResNet50-640 RetinaNet, 1/2/4/8 MI355X").

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

`value` = data-parallel TRAINING throughput (the path that shards): every rank runs the per-GPU
shard of BASELINE configs[2] — ResNet50-640x640 bf16, 32 images per GPU (global batch 256 at
N=8), `resnet_initial` frozen, SyncBN when N>1 — and one step is
  targets encoded on the GPU (a2-a4) -> forward with training-mode BatchNorm -> focal + Huber
  loss forward/backward -> backward through heads / FPN / ResNet -> weight decay, per-tensor and
  global clipping -> RCCL all-reduce of the gradient arena -> SGD momentum + EMA.
Weak scaling: per-GPU work is fixed, value = N * 32 * K / max-over-ranks time.

At N=1 the line also carries `infer`: BASELINE configs[1] (bf16 inference, batch 8, images
resident in HBM -> backbone + FPN + heads -> decode + per-class top-k + NMS -> detections),
which does not shard ("replicas only"), and `cpu_baseline`: the CPU restatement
(oracle/model_ref.py, PyTorch-CPU — NOT TensorFlow) timed on the host cores.

`roofline` = the dominant kernel of the timed region — whichever implicit-GEMM kernel (conv_halo_kernel,
conv_big_kernel, conv_fwd_kernel<128,128,64>; forward convs and every dgrad) took most time: sum of ALGORITHMIC
FLOPs of its launches (2*Ho*Wo*k*k*Cin*Cout of the layer: a dgrad launch counts its layer's MACs, not the
zero-upsampled / channel-padded GEMM it executes) / sum of their HIP-event times (events recorded on the launch
stream inside the timed steps), against 2.5 PFLOP/s dense bf16 (one entry per kernel: the device symbols that differ
only in the BN_BWD epilogue flag are summed and listed under `roofline.symbols`); the others are listed under `other_conv_kernels`,
and the HBM-bound kernels (residual 1x1 convs on conv_big_kernel<false,true,*>, the BatchNorm passes) under
`hbm_kernels` as GB/s of algorithmic bytes against 8 TB/s — in the timed (two-stream) steps and, under `exclusive`, in the
one-stream step where they do not share HBM with the weight-gradient stream; `hbm_kernels.stream_reference` is what a plain
2-read + 1-write elementwise pass reaches on the same box at the same tensor size.
"""
import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0      # HBM3E, same guide
PUBLISHED_TRAIN_IMG_S = 1290.65   # TPU v3-32, global batch 256 (README.md:75-76)


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def conv_flops(engine, step_name):
    """Algorithmic FLOPs (2*MACs) of one inference conv launch, from the static graph."""
    g = engine.g
    name = step_name.split(":", 1)[1]
    ops = [o for o in g.ops if o["op"] == "conv" and (o.get("group") == name or o["out"] == name)]
    total = 0
    for o in ops:
        c = g.convs[o["conv"]]
        Ho, Wo, Cout, _ = g.tensors[o["out"]]
        total += 2 * engine.B * Ho * Wo * c["k"] * c["k"] * c["cin"] * Cout
    return total


def conv_variant(engine, step_name):
    """Kernel the dispatcher picks for an inference conv launch (named like its device symbol): conv_halo_kernel /
    conv_big_kernel <f32 out, residual> or conv_fwd_kernel<128,128,64> (Cout > 64, Cin % 64 == 0), bf16 output;
    None for the other variants."""
    import ctypes
    g = engine.g
    p = getattr(engine, "conv_problems", {}).get(step_name)
    if p is None or step_name == "conv:stem":
        return None
    name = step_name.split(":", 1)[1]
    for o in g.ops:
        if o["op"] == "conv" and (o.get("group") == name or o["out"] == name):
            c = g.convs[o["conv"]]
            if o["out_dtype"] != "bf16":
                return None
            kid = engine.lib.rn_conv_kernel_id(ctypes.byref(p))
            res = "true" if any(p.seg[i].residual for i in range(p.num_segments)) else "false"
            if kid == 2:
                return f"conv_halo_kernel<false, {res}, false, false, 2> (256x256x32, 3x3 halo patch)"
            if kid == 3:
                return f"conv_halo_kernel<false, {res}, false, false, 4> (512x128x32, 3x3 halo patch)"
            if kid == 1:
                return f"conv_big_kernel<false, {res}, false> (256x256x32)"
            return "conv_fwd_kernel<128,128,64,bf16>" if c["cout"] > 64 and c["cin"] % 64 == 0 else None
    return None


def synth_ground_truth(B, size, seed):
    """G ~ U{1..32} boxes per image, centre U(0,size)^2, w,h = exp(U(ln 8, ln 512)),
    clipped to the image, classes U{0..79}."""
    rng = np.random.default_rng(seed)
    Gmax = 32
    gb, gc, cnt = np.zeros([B, Gmax, 4], np.float32), np.zeros([B, Gmax], np.float32), np.zeros([B], np.int32)
    for i in range(B):
        G = int(rng.integers(1, 33))
        c = rng.uniform(0, size, (G, 2))
        wh = np.exp(rng.uniform(np.log(8), np.log(512), (G, 2)))
        x1, x2 = np.clip(c - wh / 2, 0, size), np.clip(c + wh / 2, 0, size)
        gb[i, :G] = np.concatenate([(x1 + x2) / 2, np.maximum(x2 - x1, 1.0)], 1)
        gc[i, :G] = rng.integers(0, 80, G)
        cnt[i] = G
    return torch.from_numpy(gb), torch.from_numpy(gc), torch.from_numpy(cnt)


# ------------------------------------------------------------------------------------------------
def run_train(args, dev, rank, world, params=None):
    """params: another configuration than BASELINE configs[2] (the `extra` entries of the N = 1 line)"""
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    B = args.train_batch
    if params is None:
        params = default_params(input_size=args.size, batch_train=B * world)
    builder = ModelBuilder(params, "train", device=dev, seed=1337)
    model = builder()
    rx = [builder.FREEZE_VARS_REGEX[n] for n in params.training.freeze_variables]
    eng = TrainEngine(model, B, frozen_regexes=rx, world_size=world)
    if world > 1:
        from retinanet import comm
        comm.maybe_enable_native(eng)       # SyncBN / C2 messages through rn_comm (direct RCCL) when it validates
    enc = LabelEncoder(params, device=dev)
    gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(B, args.size, 1337 + rank)]
    images = torch.randn((B, args.size, args.size, 3), generator=torch.Generator().manual_seed(1337 + rank)).to(dev)

    def step():
        targets = enc.encode_batch(gb, gc, cnt)
        return eng.train_step(images, targets)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events bracket the conv launches of every 5th timed step (and the last one): two event records
    # per launch on ~90 launches cost ~1.5 ms, which would otherwise inflate every timed step
    prof, hbm, wprof, lprof = [], [], [], []
    sampled = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        sample = (i % 5 == 4) or i == args.steps - 1
        eng.conv_profile = prof if sample else None
        eng.hbm_profile = hbm if sample else None
        eng.wgrad_profile = wprof if sample else None
        eng.layer_profile = lprof if sample else None
        sampled += int(sample)
        out = step()
    eng.conv_profile = eng.hbm_profile = eng.wgrad_profile = eng.layer_profile = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    # per kernel: [ms, flops, bytes, launches]; the dominant kernel is the one with most time.  The third template flag
    # of the 256-row kernels (BN_BWD: the epilogue also writes stage 1 of a BatchNorm backward reduction) does not make
    # another kernel — same K loop, same tiles — so its two device symbols are ONE roofline entry, with the
    # per-symbol figures kept under `symbols` (what a rocprof kernel-stats row can be matched against).
    def family(variant):
        return re.sub(r"^(conv_(?:halo|big)_kernel<\w+, \w+), \w+([,>])", r"\1, *\2", variant)

    def per_kernel(events, key=family):
        acc_by = {}
        for e0, e1, fl, by, variant in events:
            acc = acc_by.setdefault(key(variant), [0.0, 0, 0, 0])
            acc[0] += e0.elapsed_time(e1); acc[1] += fl; acc[2] += by; acc[3] += 1
        return acc_by
    by_kernel = per_kernel(prof)
    by_symbol = per_kernel(prof, key=lambda v: v)
    def hbm_table(events, conv_by, steps_):
        """HBM-bound kernels: algorithmic bytes / event time against 8 TB/s (BatchNorm passes + the residual 1x1 layers)"""
        by = {}
        for e0, e1, kind, byts in events:
            acc = by.setdefault(kind + "_kernel", [0.0, 0, 0])
            acc[0] += e0.elapsed_time(e1); acc[1] += byts; acc[2] += 1
        for name in [k for k in conv_by if k.startswith("conv_big_kernel<false, true, *>")]:   # residual 1x1 layers
            v = conv_by[name]
            by[name] = [v[0], v[2], v[3]]
        return {k: {"GB/s": round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] else 0.0,
                    "frac": round(v[1] / (v[0] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if v[0] else 0.0,
                    "ms_per_step": round(v[0] / max(steps_, 1), 3), "launches_per_step": v[2] // max(steps_, 1),
                    "algorithmic_MB_per_launch": round(v[1] / max(v[2], 1) / 1e6, 2)}
                for k, v in sorted(by.items())}
    hbm_kernels = hbm_table(hbm, by_kernel, sampled)
    # the fused stage-1 bottleneck launches (rn_bottleneck64_fwd): HBM-bound by design — algorithmic bytes = block input +
    # output — and the former home of the best-running residual 1x1 layers, which therefore left the conv_big row above
    bn_ev = [(e0, e1, fl, by) for e0, e1, name, fl, by, kern in lprof if kern.startswith("bneck64")]
    if bn_ev:
        ms, byts = sum(e0.elapsed_time(e1) for e0, e1, _, _ in bn_ev), sum(b for _, _, _, b in bn_ev)
        hbm_kernels["bneck64_kernel (one launch per stage-1 bottleneck block)"] = {
            "GB/s": round(byts / (ms * 1e-3) / 1e9, 1), "frac": round(byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "ms_per_step": round(ms / max(sampled, 1), 3), "launches_per_step": len(bn_ev) // max(sampled, 1),
            "algorithmic_MB_per_launch": round(byts / len(bn_ev) / 1e6, 2)}
    dom_name = max(by_kernel, key=lambda k: by_kernel[k][0]) if by_kernel else "none"
    dom_ms, dom_flops, dom_bytes, dom_n = by_kernel.get(dom_name, [0.0, 0, 0, 0])
    # The timed steps run the weight-gradient launches on a second stream (TrainEngine.backward), so the dgrad
    # launches of the dominant kernel share the chip with wgrad kernels and their event-bracketed durations above
    # include that sharing.  One more, untimed, step with the one-stream backward gives the kernel's own duration.
    def wgrad_entry(events, steps_):
        """weight-gradient launches (rn_conv2d_nhwc_wgrad = partial-tile kernel + ordered split-K reduction): algorithmic
        FLOPs = 2 * pixels * k*k * Cin * Cout of the layer / HIP-event time on the launch stream, per kernel family"""
        by = {}
        for e0, e1, fl, name in events:
            acc = by.setdefault(name, [0.0, 0, 0])
            acc[0] += e0.elapsed_time(e1); acc[1] += fl; acc[2] += 1
        tot_ms, tot_fl = sum(v[0] for v in by.values()), sum(v[1] for v in by.values())
        if not tot_ms:
            return None
        return {"bound": "mfma", "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                "ms_per_step": round(tot_ms / max(steps_, 1), 3),
                "launches_per_step": sum(v[2] for v in by.values()) // max(steps_, 1),
                "algorithmic_gflop_per_step": round(tot_fl / max(steps_, 1) / 1e9, 1),
                "kernels": {k: {"ms_per_step": round(v[0] / max(steps_, 1), 3), "launches_per_step": v[2] // max(steps_, 1),
                                "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] else 0.0,
                                "frac": round(v[1] / (v[0] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if v[0] else 0.0}
                            for k, v in sorted(by.items())}}
    def layer_table(events, events_excl):
        """`roofline.layers`: one row per implicit-GEMM launch of the training step — forward, data
        gradient, weight gradient — from the same HIP-event samples: arithmetic intensity AI = algorithmic FLOPs /
        algorithmic bytes decides the roof (MFMA when AI >= peak FLOP/s / peak B/s = 312 FLOP/B, else HBM), `achieved` /
        `frac` are against THAT roof, in the timed two-stream steps and (`*_exclusive`) in the one-stream step."""
        ridge = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        rows, order = {}, []
        for tag, evs in (("in_step", events), ("exclusive", events_excl or [])):
            for e0, e1, name, fl, by, kern in evs:
                if name not in rows:
                    rows[name] = {"layer": name, "kernel": kern, "gflop": fl / 1e9, "MB": by / 1e6, "ms": {}, "n": {}}
                    order.append(name)
                r = rows[name]
                r["ms"][tag] = r["ms"].get(tag, 0.0) + e0.elapsed_time(e1)
                r["n"][tag] = r["n"].get(tag, 0) + 1
        out = []
        for name in order:
            r = rows[name]
            ai = r["gflop"] * 1e9 / max(r["MB"] * 1e6, 1.0)
            bound = "mfma" if ai >= ridge else "hbm"
            row = {"layer": name, "kernel": r["kernel"], "AI": round(ai, 1), "bound": bound}
            for tag, sfx in (("in_step", ""), ("exclusive", "_exclusive")):
                if tag not in r["ms"] or not r["ms"][tag]:
                    continue
                us = r["ms"][tag] * 1e3 / r["n"][tag]
                if bound == "mfma":
                    ach, peak = r["gflop"] * 1e9 / (us * 1e-6) / 1e12, PEAK_BF16_TFLOPS
                else:
                    ach, peak = r["MB"] * 1e6 / (us * 1e-6) / 1e9, PEAK_HBM_GBS
                row["us" + sfx] = round(us, 1)
                row["achieved" + sfx] = round(ach, 1)
                row["frac" + sfx] = round(ach / peak, 4)
            row["unit"] = "TFLOP/s" if bound == "mfma" else "GB/s"
            out.append(row)
        return out
    wgrad_roof = wgrad_entry(wprof, sampled)
    if wgrad_roof:
        wgrad_roof["concurrency"] = ("second stream: these launches run beside the data-gradient chain of the main stream "
                                     "in the timed steps" if getattr(eng, "side_stream_on", False) else "one stream")
    exclusive = None
    lprof1 = None
    if getattr(eng, "side_stream_on", False) and dom_name in by_kernel and not args.no_exclusive:
        eng.side_stream_on = False
        eng.set_wgrad_cap(False)     # the two-stream step caps the weight-gradient grids to leave CUs to the main stream
        prof1, wprof1, hbm1, lprof1 = [], [], [], []
        eng.conv_profile, eng.wgrad_profile, eng.hbm_profile, eng.layer_profile = prof1, wprof1, hbm1, lprof1
        step()
        eng.conv_profile = eng.wgrad_profile = eng.hbm_profile = eng.layer_profile = None
        torch.cuda.synchronize()
        eng.side_stream_on = True
        eng.set_wgrad_cap(True)
        # the HBM-bound passes with the chip (and its memory system) to themselves: in the timed steps they share HBM with
        # the second stream's weight-gradient kernels, which is not the passes' own inefficiency
        for k, v in hbm_table(hbm1, per_kernel(prof1), 1).items():
            if k in hbm_kernels:
                hbm_kernels[k]["exclusive"] = {kk: v[kk] for kk in ("GB/s", "frac", "ms_per_step")}
        w1 = wgrad_entry(wprof1, 1)
        if wgrad_roof and w1:
            wgrad_roof["exclusive"] = {k: w1[k] for k in ("achieved", "frac", "ms_per_step", "kernels")}
            wgrad_roof["exclusive"]["how"] = "the same launches in one extra untimed one-stream step (chip to themselves)"
        ms1, fl1, _, n1 = per_kernel(prof1).get(dom_name, [0.0, 0, 0, 0])
        excl_symbols = {k: {"launches": v[3], "avg_launch_us": round(v[0] * 1e3 / max(v[3], 1), 2)}
                        for k, v in per_kernel(prof1, key=lambda v: v).items() if family(k) == dom_name}
        if ms1:
            exclusive = {"achieved": round(fl1 / (ms1 * 1e-3) / 1e12, 2),
                         "frac": round(fl1 / (ms1 * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                         "avg_launch_us": round(ms1 * 1e3 / max(n1, 1), 2), "launches": n1, "symbols": excl_symbols,
                         "how": "one extra untimed step with RNET_WGRAD_STREAM=0 semantics (one-stream backward): "
                                "the same launches with the chip to themselves"}
    res = {"dt": dt, "B": B, "loss": float(out["weighted-loss"].item()),
           "grad_norm": float(out["gradient-norm"].item()),
           "roofline": {"bound": "mfma", "achieved": round(dom_flops / (dom_ms * 1e-3) / 1e12, 2) if dom_ms else 0.0,
                        "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(dom_flops / (dom_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if dom_ms else 0.0,
                        "traffic": None, "kernel": dom_name + " (forward + dgrad launches)",
                        "launches_per_step": dom_n // max(sampled, 1), "sampled_steps": sampled,
                        "avg_launch_us": round(dom_ms * 1e3 / max(dom_n, 1), 2),
                        "symbols": {k: {"launches_per_step": v[3] // max(sampled, 1),
                                        "avg_launch_us": round(v[0] * 1e3 / max(v[3], 1), 2),
                                        "algorithmic_gflop_per_launch": round(v[1] / max(v[3], 1) / 1e9, 3)}
                                    for k, v in by_symbol.items() if family(k) == dom_name},
                        "ms_per_step": round(dom_ms / max(sampled, 1), 3),
                        "algorithmic_gflop_per_launch": round(dom_flops / max(dom_n, 1) / 1e9, 3),
                        "algorithmic_bytes_per_launch": int(dom_bytes / max(dom_n, 1)),
                        "concurrency": ("dgrad launches overlap wgrad launches of a second stream in the timed steps"
                                        if getattr(eng, "side_stream_on", False) else "one stream"),
                        "exclusive": exclusive,
                        "wgrad": wgrad_roof,
                        "layers": layer_table(lprof, lprof1),
                        "hbm_kernels": {"bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "kernels": hbm_kernels,
                                        "stream_reference": stream_reference(dev)},
                        "other_conv_kernels": {k: {"ms_per_step": round(v[0] / max(sampled, 1), 3),
                                                   "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] else 0.0}
                                               for k, v in by_kernel.items() if k != dom_name}}}
    return res, params, model, eng


def stream_reference(dev):
    """What a plain elementwise pass reaches on THIS box at the step's tensor size: torch.add of two 419 MB bf16 tensors
    (ResNet stage 1, B = 32: 2 reads + 1 write — the shape of bn_bwd_apply).  Context for `hbm_kernels`: the 8 TB/s peak is
    not reachable by a streaming kernel (MI355X_MICROARCH.md: 6.3 TB/s float4 copy; 2R + 1W at this size: 4.8-5.5 TB/s,
    tools/probes/stream_probe.hip)."""
    n = 32 * 160 * 160 * 256
    a = torch.ones((n,), dtype=torch.bfloat16, device=dev)
    b = torch.ones((n,), dtype=torch.bfloat16, device=dev)
    o = torch.empty_like(a)
    for _ in range(3):
        torch.add(a, b, out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.add(a, b, out=o)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gbs = 3 * n * 2 / (ms * 1e-3) / 1e9
    return {"GB/s": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4),
            "what": "torch.add(a, b, out=o), three 419 MB bf16 tensors (2 reads + 1 write), 10 launches"}


def mfma_sustained(dev):
    """What the matrix pipe of THIS part sustains, measured live (csrc/rn_probe.hip: nothing but v_mfma_f32_32x32x16_bf16,
    2 waves per SIMD, no memory traffic): on random operands and on zeros, with the core clock workgroup 0 saw
    (shader-clock ticks / 100 MHz wall-clock ticks).  The chip clocks to its power budget, so the nominal 2.5 PFLOP/s
    (2.4 GHz) is not reachable on real data; `frac_of_sustained` prices a kernel against the random-operand figure."""
    from retinanet import _C
    lib = _C.lib()
    iters = 4096
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    out = torch.empty((4 * cus * 512,), dtype=torch.float32, device=dev)
    clocks = torch.zeros((4,), dtype=torch.int64, device=dev)
    res = {}
    for name, table in (("random", torch.rand((1024,), generator=torch.Generator().manual_seed(7)).sub_(0.5).to(dev)),
                        ("zeros", torch.zeros((1024,), device=dev))):
        best, ghz = 0.0, 0.0
        for rep_ in range(4):    # the first launch warms up; the clock settles within a launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _C.check(lib.rn_probe_mfma(_C.ptr(table), _C.ptr(out), iters, _C.ptr(clocks), _C.current_stream()), "rn_probe_mfma")
            e1.record()
            torch.cuda.synchronize()
            tf = lib.rn_probe_mfma_flops(iters) / (e0.elapsed_time(e1) * 1e-3) / 1e12
            c = clocks.tolist()
            if rep_ and tf > best:
                best, ghz = tf, (c[1] - c[0]) / max(c[3] - c[2], 1) * 0.1      # ticks per 10 ns -> GHz
        res[name] = {"tflops": round(best, 1), "core_clock_ghz": round(ghz, 3)}
    res["how"] = ("rn_probe_mfma: 16 x v_mfma_f32_32x32x16_bf16 per wave and iteration, 8 waves per workgroup, no memory "
                  "traffic; best of 3 launches; core clock = s_memtime ticks / wall_clock64 (100 MHz) ticks of workgroup 0")
    return res


def run_infer(args, dev, rank):
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    B = args.infer_batch
    params = default_params(input_size=args.size, inference_batch=B)
    builder = ModelBuilder(params, "val", device=dev, seed=1337)
    model = builder()
    images = torch.randn((B, args.size, args.size, 3), generator=torch.Generator().manual_seed(1337 + rank)).to(dev)
    if args.logit_std > 0:
        # rescale the class prediction kernel so logits ~ N(-4.595, std): with the raw initialiser every
        # score is 0.01 < 0.05 and the NMS stage would have no work (SURVEY §8(d) microbench distribution)
        preds = model(images)
        std = torch.cat([preds["class-predictions"][l].reshape(-1) for l in "34567"]).std().item()
        model.variables["class-head/class-head-prediction-conv2d/kernel"].mul_(args.logit_std / max(std, 1e-12))
        model._refresh()
    infer = builder.add_post_processing_stage(model)
    engine, post = model.inference_engine(B), infer.post
    from retinanet import _C
    variants = {i: conv_variant(engine, n) for i, (_, n) in enumerate(engine.steps)}
    dom = set(i for i, v in variants.items() if v)
    ev = []

    def step(record):
        st = _C.current_stream()
        for i, (fn, _) in enumerate(engine.steps):
            if record and i in dom:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(st)
                e1.record()
                ev.append((e0, e1, i))
            else:
                fn(st)
        return post(engine.outputs)
    engine.t["images"].copy_(images)
    # timed region: what a deployment runs — `serving_default` as a HIP-graph replay (forward on two streams + the
    # post-processing stage, one graph launch per batch; retinanet/model/builder.py::add_post_processing_stage)
    infer_g = builder.add_post_processing_stage(model, capture_graph=True)
    for _ in range(max(args.warmup, 2)):
        out = infer_g(engine.t["images"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.infer_steps):
        out = infer_g(engine.t["images"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {k: v.clone() for k, v in out.items()}
    # roofline attribution, outside the timed region: eager one-stream steps with HIP events around every conv launch
    sampled = max(args.infer_steps // 5, 1)
    step(False)
    for k in range(sampled):
        step(True)
    torch.cuda.synchronize()
    by_kernel = {}
    for a, b, i in ev:
        acc = by_kernel.setdefault(variants[i], [0.0, 0, 0])
        acc[0] += a.elapsed_time(b); acc[1] += conv_flops(engine, engine.steps[i][1]); acc[2] += 1
    dom_name = max(by_kernel, key=lambda k: by_kernel[k][0]) if by_kernel else "none"
    dom_ms, dom_fl, dom_n = by_kernel.get(dom_name, [0.0, 0, 0])
    ach = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms else 0.0
    res = {"workload": f"ResNet50-{args.size}x{args.size} bf16 inference batch={B} (BASELINE configs[1]): forward + decode "
                       "+ per-class top-k 5000 + per-class NMS as one HIP-graph replay per batch; replicas only",
           "value": round(B * args.infer_steps / dt, 2), "unit": "images/s", "ms_per_step": round(dt / args.infer_steps * 1e3, 3),
           "steps": args.infer_steps, "valid_detections": out["valid_detections"].tolist(),
           "data": "synthetic N(0,1) images, reference initialisers"
                   + (f", class logits rescaled to std {args.logit_std}" if args.logit_std > 0 else ""),
           "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_BF16_TFLOPS, 4), "kernel": dom_name,
                        "launches_per_step": dom_n // max(sampled, 1), "sampled_steps": sampled,
                        "ms_per_step": round(dom_ms / max(sampled, 1), 3),
                        "other_conv_kernels": {k: {"ms_per_step": round(v[0] / max(sampled, 1), 3),
                                                   "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] else 0.0}
                                               for k, v in by_kernel.items() if k != dom_name}}}
    return res, params, model


def run_infer_b1(args, dev):
    """BASELINE configs[0] on the HIP path — the reference's own inference protocol (README.md:31-32 publishes only batch-1
    latencies; evaluate_saved_model.py:60-72 times `serving_fn(image=...)` per image and averages 1 / dt with an
    AverageMeter of momentum 0.975): ONE 640 x 640 image resident in HBM -> `serving_default` as a HIP-graph replay ->
    detections, host wall clock around each call with a device synchronisation inside (what a caller that reads the
    detections waits for).  5 warm-up calls, then `--b1-calls` timed ones: median, EMA the reference's way, and the
    back-to-back rate without a host read in between.  One eager pass with HIP events around every launch names the
    kernel that takes most of the image's time."""
    from retinanet import _C
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    params = default_params(input_size=args.size, inference_batch=1)
    builder = ModelBuilder(params, "val", device=dev, seed=1337)
    model = builder()
    image = torch.randn((1, args.size, args.size, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
    if args.logit_std > 0:   # as run_infer: logits ~ N(-4.595, std) so that NMS has the work a trained detector gives it
        preds = model(image)
        std = torch.cat([preds["class-predictions"][l].reshape(-1) for l in "34567"]).std().item()
        model.variables["class-head/class-head-prediction-conv2d/kernel"].mul_(args.logit_std / max(std, 1e-12))
        model._refresh()
    # eager pass: per-launch HIP events (the serving stage's launches as one bracket)
    infer_e = builder.add_post_processing_stage(model)
    engine, post = model.inference_engine(1), infer_e.post
    st = _C.current_stream()
    engine.t["images"].copy_(image)
    for _ in range(3):
        for fn, _n in engine.steps:
            fn(st)
        post(engine.outputs)
    torch.cuda.synchronize()
    acc, reps = {}, 10
    for _ in range(reps):
        evs = []
        for i, (fn, name) in enumerate(engine.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(st); e1.record()
            evs.append((i, name, e0, e1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); post(engine.outputs); e1.record()
        evs.append((len(engine.steps), "post_processing", e0, e1))
        torch.cuda.synchronize()
        for i, name, a, b in evs:
            acc[(i, name)] = acc.get((i, name), 0.0) + a.elapsed_time(b) / reps
    fam = {}
    for (i, name), ms in acc.items():
        v = conv_variant(engine, name) if name.startswith("conv:") else None
        key = v or ("conv_fwd_kernel (other shapes) / stem" if name.startswith("conv:") else name.split(":")[0])
        f = fam.setdefault(key, [0.0, 0, 0])
        f[0] += ms; f[1] += conv_flops(engine, name) if v else 0; f[2] += 1
    dom = max((k for k in fam if fam[k][1] > 0), key=lambda k: fam[k][0])
    # the protocol itself: graph replay of serving_default
    infer = builder.add_post_processing_stage(model, capture_graph=True)
    for _ in range(5):
        out = infer(image)
    torch.cuda.synchronize()
    times, fps_ema = [], None
    for k in range(args.b1_calls):
        t0 = time.perf_counter()
        out = infer(image)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        times.append(dt)
        fps_ema = (1.0 / dt) if fps_ema is None else fps_ema * 0.975 + 0.025 / dt    # AverageMeter(momentum=0.975) of 1 / dt
    med = float(np.median(times))
    n = args.b1_calls
    t0 = time.perf_counter()
    for _ in range(n):
        out = infer(image)
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / n
    launches = len(engine.steps) + getattr(infer_e, "launches_per_call", 0)
    ach = fam[dom][1] / (fam[dom][0] * 1e-3) / 1e12 if fam[dom][0] else 0.0
    return {"workload": f"ResNet50-{args.size}x{args.size} bf16 inference batch=1 (BASELINE configs[0]; the reference's protocol, "
                        "evaluate_saved_model.py:60-72): HIP-graph replay of serving_default = forward + decode + per-class "
                        "top-k 5000 + per-class NMS, image resident in HBM, host wall clock per call incl. the device sync",
            "value": round(1.0 / med, 2), "unit": "images/s", "median_ms": round(med * 1e3, 4),
            "ema_ms": round(1e3 / fps_ema, 4), "ema_fps": round(fps_ema, 2), "p90_ms": round(float(np.percentile(times, 90)) * 1e3, 4),
            "back_to_back_ms": round(b2b * 1e3, 4), "timed_calls": n, "warmup_calls": 5,
            "valid_detections": out["valid_detections"].tolist(),
            "engine_launches_per_image": len(engine.steps),
            "eager_event_ms_per_image": round(sum(acc.values()), 4),
            "published_context": {"TF-TensorRT FP16, Tesla V100 (README.md:31)": "11.0 ms (90.1 FPS)",
                                  "TF FP32, Tesla V100": "25.0 ms (40.5 FPS)",
                                  "note": "other hardware, trained weights, TensorRT engine: context, not a baseline for vs_baseline"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_BF16_TFLOPS, 4), "launches_per_image": fam[dom][2],
                         "ms_per_image": round(fam[dom][0], 4),
                         "by_kernel_ms": {k: {"ms": round(v[0], 4), "launches": v[2]} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}}}


def run_dp_overhead(args, dev):
    """What the data-parallel machinery costs on ONE GPU (a multi-GPU node is not available to the
    builder): the B = 32 training step on a 1-rank `nccl` process group with the DP path forced on — SyncBatchNorm messages
    through torch.distributed (RCCL), ~25 MB gradient buckets prepared and all-reduced on the third stream as the backward
    pass completes them, the clip-flag host read — against the plain single-replica step: two engines in ONE process,
    alternating rounds.  `without_flag_read` prices the one host read per step (`TrainEngine._overlap_finish`)."""
    import socket
    from retinanet.cfg import default_params
    from retinanet.dataloader import LabelEncoder
    from retinanet.model import ModelBuilder
    from retinanet.model.train_engine import TrainEngine
    own_pg = not dist.is_initialized()
    if own_pg:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        B = args.train_batch
        params = default_params(input_size=args.size, batch_train=B)
        builder = ModelBuilder(params, "train", device=dev, seed=1337)
        model = builder()
        rx = [builder.FREEZE_VARS_REGEX[n] for n in params.training.freeze_variables]
        enc = LabelEncoder(params, device=dev)
        gb, gc, cnt = [t.to(dev) for t in synth_ground_truth(B, args.size, 1337)]
        images = torch.randn((B, args.size, args.size, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
        engs = {"plain": TrainEngine(model, B, frozen_regexes=rx, world_size=1, force_dp=False),
                "dp": TrainEngine(model, B, frozen_regexes=rx, world_size=1, force_dp=True)}

        def run(name, n, no_read=False, overlap=None):
            eng = engs[name]
            eng.price_without_flag_read = no_read
            old = os.environ.get("RNET_C1_OVERLAP")
            if overlap is not None:
                os.environ["RNET_C1_OVERLAP"] = overlap        # read at the head of every backward pass
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    out = eng.train_step(images, enc.encode_batch(gb, gc, cnt))
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3, out
            finally:
                if overlap is not None:
                    os.environ.pop("RNET_C1_OVERLAP", None)
                    if old is not None:
                        os.environ["RNET_C1_OVERLAP"] = old
        for name in engs:
            run(name, 3)
        rounds = {"plain": [], "dp": [], "dp_without_flag_read": [], "dp_plain_order": []}
        for _ in range(4):
            rounds["plain"].append(run("plain", 5)[0])
            ms, out = run("dp", 5)
            rounds["dp"].append(ms)
            rounds["dp_without_flag_read"].append(run("dp", 5, no_read=True)[0])
            rounds["dp_plain_order"].append(run("dp", 5, overlap="0")[0])
        med = {k: float(np.median(v)) for k, v in rounds.items()}
        eng = engs["dp"]
        return {"workload": f"ResNet50-{args.size} bf16 training step, {B} images, ONE GPU: plain step vs the same step with the "
                            "data-parallel path forced on over a 1-rank nccl (RCCL) group — SyncBN messages, bucketed gradient "
                            "all-reduce overlapped with the backward pass (RNET_C1_OVERLAP default), clip-flag host read",
                "ms_per_step": round(med["dp"] - med["plain"], 3), "plain_ms": round(med["plain"], 3), "dp_ms": round(med["dp"], 3),
                "dp_without_flag_read_ms": round(med["dp_without_flag_read"], 3),
                "dp_plain_order_ms": round(med["dp_plain_order"], 3),     # RNET_C1_OVERLAP=0: all-reduce after the backward pass
                "overlap_enabled": not bool(getattr(engs["dp"], "_overlap_unsafe", False)),
                "max_host_ms_in_a_bucket_all_reduce_call": round(float(getattr(engs["dp"], "bucket_host_ms", 0.0)), 3),
                "flag_read_ms": round(med["dp"] - med["dp_without_flag_read"], 3),
                "syncbn_messages_per_step": eng.syncbn_messages_per_step, "gradient_buckets": len(getattr(eng, "_buckets", []) or []),
                "clip_fired_last_step": bool(getattr(eng, "clip_fired", False)),
                "rounds_ms": {k: [round(x, 3) for x in v] for k, v in rounds.items()},
                "protocol": "two engines in one process, 3 warm-up steps each, 4 alternating rounds of 5 steps, medians",
                "note": "one rank: RCCL moves no bytes over xGMI; this is the enqueue / kernel / stream-hop cost of the machinery"}
    finally:
        if own_pg:
            dist.destroy_process_group()


def run_extras(args, dev):
    """Driver-visible numbers for BASELINE configs[3] and configs[4] on one GPU: a few training steps of
    ResNet50-1024x1024 (bf16, 16 images: the HBM-bound FPN path) and of EfficientNet-B3 640x640 under its own policy
    (`mixed_float16`: IEEE-half storage on librnet_hip_f16.so, dynamic loss scale), plus EfficientNet-B3 batch-8 inference
    with soft-NMS.  Each with its dominant bracketed kernel: MFMA TFLOP/s for config 3, HBM GB/s of algorithmic bytes for
    config 4 (its 1x1 convs, depthwise convs and BatchNorm passes are HBM-bound)."""
    import copy
    from retinanet.cfg import default_params, efficientnet_params
    from retinanet.model import ModelBuilder
    extra = {}
    a3 = copy.copy(args)
    a3.size, a3.train_batch, a3.steps, a3.warmup, a3.no_exclusive = 1024, 16, 5, 2, True
    r3, _, _, eng = run_train(a3, dev, 0, 1, default_params(input_size=1024, batch_train=16))
    rl = r3["roofline"]
    extra["config3"] = {"workload": "ResNet50-1024x1024 bf16 training, 16 images on one GPU (BASELINE configs[3] shard), "
                                    "resnet_initial frozen", "value": round(16 * a3.steps / r3["dt"], 2), "unit": "images/s",
                        "ms_per_step": round(r3["dt"] / a3.steps * 1e3, 3), "steps": a3.steps, "warmup": a3.warmup,
                        "final_loss": round(r3["loss"], 4),
                        "roofline": {k: rl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches_per_step",
                                                        "avg_launch_us", "ms_per_step", "algorithmic_gflop_per_launch")}}
    del eng, r3
    torch.cuda.empty_cache()
    p4 = efficientnet_params("efficientnet-b3", input_size=640)
    p4.architecture.batch_norm.use_sync = False
    a4 = copy.copy(args)
    a4.size, a4.train_batch, a4.steps, a4.warmup, a4.no_exclusive = 640, 32, 5, 2, True
    r4, _, _, eng = run_train(a4, dev, 0, 1, p4)
    hb = r4["roofline"]["hbm_kernels"]["kernels"]
    conv = dict(r4["roofline"].get("other_conv_kernels", {}))
    dom = max(hb, key=lambda k: hb[k]["ms_per_step"]) if hb else None
    extra["config4"] = {"workload": "EfficientNet-B3 640x640 training, 32 images on one GPU (BASELINE configs[4] shard), "
                                    f"policy {p4.floatx.precision}: IEEE-half storage on librnet_hip_f16.so, dynamic loss scale",
                        "value": round(32 * a4.steps / r4["dt"], 2), "unit": "images/s", "dtype": "f16",
                        "ms_per_step": round(r4["dt"] / a4.steps * 1e3, 3), "steps": a4.steps, "warmup": a4.warmup,
                        "final_loss": round(r4["loss"], 4),
                        "roofline": ({"bound": "hbm", "achieved": hb[dom]["GB/s"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": hb[dom]["frac"], "kernel": dom, "ms_per_step": hb[dom]["ms_per_step"],
                                      "launches_per_step": hb[dom]["launches_per_step"],
                                      "algorithmic_MB_per_launch": hb[dom]["algorithmic_MB_per_launch"],
                                      "other_hbm_kernels": {k: {kk: v[kk] for kk in ("GB/s", "ms_per_step")}
                                                            for k, v in hb.items() if k != dom},
                                      "mfma_kernels": {r4["roofline"]["kernel"]: {"ms_per_step": r4["roofline"]["ms_per_step"],
                                                                                   "tflops": r4["roofline"]["achieved"]}, **conv}}
                                     if dom else None)}
    del eng, r4
    torch.cuda.empty_cache()
    extra["config4"]["infer"] = run_config4_infer(args, dev, p4)
    return extra


def build_config4_serving(args, dev, p4=None):
    """-> (params, ModelBuilder, model, input batch) of the configs[4] serving leg"""
    from retinanet.cfg import efficientnet_params
    from retinanet.model import ModelBuilder
    if p4 is None:
        p4 = efficientnet_params("efficientnet-b3", input_size=640)
        p4.architecture.batch_norm.use_sync = False
    bi = ModelBuilder(p4, "val", device=dev, seed=1337)
    mi = bi()
    x = torch.randn((args.infer_batch, 640, 640, 3), generator=torch.Generator().manual_seed(1337)).to(dev)
    if args.logit_std > 0:
        # At the reference's initialisation this network returns bias-only logits in inference mode: the last BatchNorm of
        # every MBConv block starts with gamma = 0, and the blocks without a skip connection then output zeros.  Give the
        # BatchNorm layers non-degenerate parameters (as the parity tests do) and then, as in run_infer, rescale the class
        # prediction kernel so that logits ~ N(-4.595, std): otherwise no score passes the threshold and NMS has no work.
        gen = torch.Generator().manual_seed(1337)
        for k, v in mi.variables.items():
            if k.endswith("/gamma") or k.endswith("/moving_variance"):
                v.copy_((torch.rand(v.shape, generator=gen) * 0.5 + 0.75).to(v.device))
            elif k.endswith("/beta") or k.endswith("/moving_mean"):
                v.copy_((torch.randn(v.shape, generator=gen) * 0.1).to(v.device))
        mi._refresh()
        preds = mi(x)
        std = torch.cat([preds["class-predictions"][l].reshape(-1) for l in "34567"]).float().std().item()
        if std > 0 and std == std:
            name = "class-head/class-head-prediction-conv2d/"
            key = name + ("pointwise_kernel" if name + "pointwise_kernel" in mi.variables else "kernel")
            mi.variables[key].mul_(args.logit_std / std)
            mi._refresh()
    return p4, bi, mi, x


def run_config4_infer(args, dev, p4=None):
    """The serving leg of BASELINE configs[4]: EfficientNet-B3 640x640, PerClassSoftNMS, one HIP-graph replay per batch."""
    p4, bi, mi, x = build_config4_serving(args, dev, p4)
    infer = bi.add_post_processing_stage(mi, capture_graph=True)
    for _ in range(3):
        infer(x)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        out = infer(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"workload": f"EfficientNet-B3 640x640 inference batch={args.infer_batch}, {p4.inference.mode} "
                        "(HIP-graph replay), random BatchNorm parameters, class logits rescaled as for `infer`",
            "value": round(args.infer_batch / dt, 2), "unit": "images/s", "ms_per_step": round(dt * 1e3, 3), "steps": n,
            "valid_detections": out["valid_detections"].tolist()}


def cpu_baseline(params_train, model_train, frozen, params_infer, model_infer, budget_s=75.0):
    """the CPU restatement (oracle/, PyTorch-CPU fp32 — NOT TensorFlow) on the host cores.
    Inference = BASELINE configs[0] (ResNet50-640, batch 1, forward + decode + per-class top-k 5000 + PerClassHardNMS)
    with the protocol of evaluate_saved_model.py:64-72: 5 warm-up calls, then timed calls (>= 50 unless the time budget
    runs out first — the count is reported), median and EMA(0.975) per image, plus a 1-thread figure.  Training = one
    warm-up step and >= 3 timed steps of batch 2 (target encode + forward + loss + autograd backward + clip + SGD)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as o
    from model_ref import RefModel, RefTrainer
    cores = usable_cores()
    torch.set_num_threads(cores)
    size = params_train.input.input_shape[0]
    ap = params_train.anchor_params
    an = o.generate_anchors(size, size, 3, 7, ap.areas, ap.aspect_ratios, ap.scales)
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    out = {"unit": "images/s", "cores": cores, "cpu": cpu_model, "kind": "port",
           "what": "CPU restatement (oracle/model_ref.py + oracle/oracle.py, PyTorch-CPU fp32) — not TensorFlow"}
    img = torch.randn((2, size, size, 3), generator=torch.Generator().manual_seed(1337))
    t_start = time.perf_counter()
    if model_infer is not None:
        ref = RefModel(params_infer, model_infer.variables, emulate_bf16=False)

        def one():
            p = ref(img[:1])
            logits = np.concatenate([p["class-predictions"][l].numpy().reshape(1, -1, 80) for l in "34567"], 1)
            encd = np.concatenate([p["box-predictions"][l].numpy().reshape(1, -1, 4) for l in "34567"], 1)
            return o.postprocess(logits, encd, an, size, size)
        for _ in range(5):
            one()
        times, ema = [], None
        while len(times) < 50 and (len(times) < 5 or time.perf_counter() - t_start < 0.45 * budget_s):
            t0 = time.perf_counter()
            one()
            dt = time.perf_counter() - t0
            times.append(dt)
            ema = dt if len(times) <= 10 else ema * 0.975 + 0.025 * dt      # AverageMeter(momentum=0.975)
        med = float(np.median(times))
        out["infer"] = {"value": round(1.0 / med, 3), "unit": "images/s", "median_ms": round(med * 1e3, 2),
                        "ema_ms": round(ema * 1e3, 2), "timed_calls": len(times), "warmup_calls": 5,
                        "sample": f"BASELINE configs[0]: ResNet50-{size}, batch 1, forward + decode + top-k 5000 + "
                                  "per-class NMS, synthetic N(0,1) image, logits at the reference's initialisation"}
        torch.set_num_threads(1)
        one()
        t1 = []
        for _ in range(2):
            t0 = time.perf_counter()
            one()
            t1.append(time.perf_counter() - t0)
        torch.set_num_threads(cores)
        out["infer"]["one_thread"] = {"value": round(1.0 / float(np.median(t1)), 3), "timed_calls": 2}
    B = 2
    gb, gc, cnt = synth_ground_truth(B, size, 1337)
    tr = RefTrainer(params_train, model_train.variables, frozen_names=frozen, dtype=torch.float32)

    def train_step():
        enc = [o.encode_sample(an, gb[i, :cnt[i]].numpy(), gc[i, :cnt[i]].numpy()) for i in range(B)]   # part of the step
        for t in tr.leaf.values():
            t.grad = None
        tr.step(img, np.stack([e[1] for e in enc]), np.stack([e[2] for e in enc]), float(sum(e[3] for e in enc)), 0.01)
    train_step()
    tt = []
    for _ in range(3):
        t0 = time.perf_counter()
        train_step()
        tt.append(time.perf_counter() - t0)
    out["value"] = round(B / float(np.median(tt)), 3)
    out["sample"] = (f"{len(tt)} timed training steps after 1 warm-up, batch {B}, ResNet50-{size}: target encode + forward "
                     "+ loss + autograd backward + clip + SGD (median)")
    return out


def launch_ranks(n):
    """Start `n` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as a
    child process and return its exit status.  Called before anything in this process has initialised the GPU."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--train-batch", type=int, default=32, help="images per GPU per training step (configs[2]: 256/8)")
    ap.add_argument("--infer-batch", type=int, default=8)
    ap.add_argument("--infer-steps", type=int, default=30)
    ap.add_argument("--b1-calls", type=int, default=200, help="timed calls of the batch-1 serving leg (infer_b1)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--logit-std", type=float, default=1.0)
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--no-exclusive", action="store_true",
                    help="skip the one extra one-stream step behind `roofline.exclusive` (profile runs: the kernel "
                         "trace then holds warm-up + timed steps only, like the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip `extra`: BASELINE configs[3] / [4] on one GPU")
    ap.add_argument("--no-probe", action="store_true", help="skip the MFMA-only probe behind roofline.sustained")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed by hand: this process becomes the LAUNCHER (distribute.py:16-18: one replica
        # per GPU).  It has not touched the GPU (importing torch does not), starts the N ranks as a CHILD process group
        # — no exec, no in-process retry —, lets rank 0's JSON line through on the inherited stdout and exits with the
        # child's status.
        sys.exit(launch_ranks(args.gpus))
    # The contract is ONE JSON line on stdout.  Native libraries write there too (librccl prints its version banner
    # through C stdio when a communicator comes up, flushed at exit — after the line): keep a private handle on the real
    # stdout for the line and point file descriptor 1 at stderr for everything else in this process.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` or plain `python bench.py "
                         f"--gpus {args.gpus}`")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RNET_BENCH_ONE_DEVICE=1: functional check of the N>1 code path on a single-GPU box — every rank uses
    # cuda:0 and the collectives go through gloo (RCCL refuses two ranks on one device).  Not a measurement.
    one_device = os.environ.get("RNET_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    train, p_train, m_train, eng = run_train(args, dev, rank, world)
    B = train["B"]
    line = {
        "metric": "images/sec train, ResNet50-640 RetinaNet data-parallel step (encode + fwd + loss + bwd + clip + "
                  "all-reduce + SGD/EMA); inference images/s under `infer`",
        "value": round(world * B * args.steps / train["dt"], 2), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(train["dt"] / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak",
        # BASELINE.md publishes ONE number for this metric: 1290.65 images/s at global batch 256 (TPU v3-32).  It is the
        # same workload only when the job's global batch is 256, i.e. at N = 8 with 32 images per GPU.
        "vs_baseline": round(world * B * args.steps / train["dt"] / PUBLISHED_TRAIN_IMG_S, 3) if world * B == 256 else None,
        "dtype": "bf16",
        "data": "synthetic N(0,1) images + U{1..32} random boxes per image (seed 1337+rank), reference initialisers",
        "config": {"workload": f"ResNet50-{args.size}x{args.size} bf16 training, {B} images/GPU (BASELINE configs[2] "
                               f"shard: global batch {world * B}), resnet_initial frozen, "
                               f"{'SyncBN + RCCL gradient all-reduce' if world > 1 else 'single GPU'}",
                   "global_batch": world * B, "parallelism": f"dp{world}",
                   # ranks of the process group the gradient all-reduce runs on, its backend ("nccl" = RCCL over xGMI;
                   # "gloo" only in the one-device functional mode) and who carries the SyncBN / normaliser messages
                   "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                   "backend": dist.get_backend() if world > 1 else None,
                   "comm": "native" if getattr(eng, "native_comm", None) is not None else "torch",
                   "syncbn_messages": getattr(eng, "syncbn_messages_per_step", None),
                   "final_loss": round(train["loss"], 4), "gradient_norm": round(train["grad_norm"], 4)},
        "roofline": train["roofline"],
    }
    if rank == 0 and not args.no_probe:
        try:
            sus = mfma_sustained(dev)
            rl = line["roofline"]
            rl["sustained"] = sus
            if sus["random"]["tflops"] > 0:
                rl["frac_of_sustained"] = round(rl["achieved"] / sus["random"]["tflops"], 4)
                if rl.get("exclusive"):
                    rl["exclusive"]["frac_of_sustained"] = round(rl["exclusive"]["achieved"] / sus["random"]["tflops"], 4)
                if rl.get("wgrad"):
                    rl["wgrad"]["frac_of_sustained"] = round(rl["wgrad"]["achieved"] / sus["random"]["tflops"], 4)
        except Exception as e:   # noqa: BLE001 — a probe failure must not lose the bench line
            line["roofline"]["sustained"] = {"error": str(e)}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            # profiles/traffic.json (tools/summarize_profiles.py): per device symbol, HBM bytes per launch from the
            # FETCH_SIZE / WRITE_SIZE passes of the same bench command
            kern = json.load(open(tpath)).get("kernels", {})
            tot = n = 0
            for sym, st in line["roofline"].get("symbols", {}).items():   # launch-weighted mean over the device symbols
                ent = kern.get(sym.split(" (")[0])
                if ent:
                    tot += ent["bytes_per_launch"] * st["launches_per_step"]
                    n += st["launches_per_step"]
            line["roofline"]["traffic"] = int(tot / n) if n else None
        except Exception:
            pass
    if world == 1 and rank == 0:
        frozen = set(eng.frozen)
        del eng
        torch.cuda.empty_cache()
        p_inf = m_inf = None
        if not args.no_infer:
            inf, p_inf, m_inf = run_infer(args, dev, rank)
            line["infer"] = inf
            try:
                line["infer_b1"] = run_infer_b1(args, dev)
            except Exception as e:   # noqa: BLE001 — the headline line must survive a failure in a side measurement
                line["infer_b1"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_extras:
            try:
                line["extra"] = run_extras(args, dev)
            except Exception as e:   # noqa: BLE001 — the headline line must survive a failure in the side measurements
                line["extra"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                line["extra"]["dp_overhead"] = run_dp_overhead(args, dev)
            except Exception as e:   # noqa: BLE001
                line["extra"]["dp_overhead"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(p_train, m_train, frozen, p_inf, m_inf)
    if rank == 0:
        line_out.write(json.dumps(line) + "\n")
        line_out.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
