"""Executors that turn the static layer graph into HIP launches.

`InferenceEngine` is the MI355X replacement of `model(images, training=False)` for the Keras
model the reference builds (retinanet/model/builder.py:94-106): weights are packed once to the
MFMA-friendly bf16 [Cout][R][S][Cin] layout, BatchNorm (inference mode: moving statistics) and
conv bias are applied in the conv epilogue together with the activation and the residual add (rounded to bf16
where the reference holds a bf16 tensor between two layers: rnet_hip.h, rn_conv_segment), all activations live in buffers allocated once, and the
launch list is fixed — so the whole forward pass can be captured in a HIP graph
(`capture_graph=True`) and replayed with one host call instead of ~70.
"""
from __future__ import annotations

import ctypes
import os

import torch

from retinanet import _C

_DT = {"bf16": torch.bfloat16, "f32": torch.float32}


def _conv_out_hw(H, W, k, s, pad):
    return (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1


def split_by_depth(g, ops):
    """A grouped forward launch whose segments differ in K depth by 2x or more (the FPN lateral 1x1 convs: 512 / 1024 /
    2048 input channels) as two launches, the deep segments first.  The persistent kernels hand every XCD a contiguous range
    of tiles, so in one launch a single XCD ends up with all 50 of the 64-step tiles and most of the 32-step ones (181 K
    steps per CU there against 87 on average: 229 us for 103 us of work at B = 32).  On their own the deep segments are
    one round of tiles; the rest is a homogeneous launch.  Forward launches without BatchNorm only (a group with live
    BatchNorm shares one statistics message); RNET_GROUP_SPLIT=0 keeps one launch (A/B)."""
    if len(ops) < 2 or os.environ.get("RNET_GROUP_SPLIT", "1") == "0":
        return [ops]
    depth = [g.convs[o["conv"]]["k"] ** 2 * g.convs[o["conv"]]["cin"] for o in ops]
    deep = [o for o, d in zip(ops, depth) if d >= 2 * min(depth)]
    rest = [o for o, d in zip(ops, depth) if d < 2 * min(depth)]
    return [deep, rest] if deep and rest else [ops]


def pixel_pair_kernel(w):
    """HWIO [3, 3, C, C] kernel of a 3x3 / stride-1 convolution -> the [3, 3, 2C, 2C] kernel of the SAME convolution over
    pixel pairs: two horizontally adjacent pixels of the NHWC tensor seen as one pixel of 2C channels ([N, H, W, C] and
    [N, H, W/2, 2C] are the same bytes).  Output pixel 2X + a (a = 0, 1), tap s reads input pixel 2X + a + s - 1 =
    2(X + S - 1) + b: pair column S = (a + s - 1) // 2 + 1, half b = (a + s - 1) % 2; every other entry is zero."""
    import torch
    C = w.shape[2]
    out = torch.zeros((3, 3, 2 * C, 2 * C), dtype=w.dtype, device=w.device)
    for a in (0, 1):
        for s_ in (0, 1, 2):
            t = a + s_ - 1
            S, b = t // 2 + 1, t % 2
            out[:, S, b * C:(b + 1) * C, a * C:(a + 1) * C] = w[:, s_]
    return out


def pixel_pair_ok(lib, g, op, B, opts, splitk_ws=None):
    """Does conv `op` run in pixel-pair form?  The 64-channel 3x3 layers of ResNet stage 1 (resnet.py:236-239 at 160 x 160)
    fill half a 128-column tile of every MFMA kernel here; on the 128-row kernel, which stages the pixels once per tap,
    they ran at 2.9x their HBM time (round 4: 120 - 131 us against 43 at B = 32).  As a convolution over pixel pairs the
    layer is 128 -> 128 channels on half as many pixels — the shape the halo kernel's 512 x 128 tiles take: twice the MACs
    (half of the paired kernel is zeros), one staging of the pixels per channel chunk.  Same products in the same order
    per output, so the same values.  Only where that kernel takes the paired shape (rn_conv_kernel_id == 3: enough tiles)
    and the layer runs in inference form (frozen `resnet_initial` layers in training, every layer when serving).
    RNET_PIXEL_PAIR=0 keeps the plain form."""
    import ctypes
    if os.environ.get("RNET_PIXEL_PAIR", "1") == "0" or op.get("op") != "conv":
        return False
    c = g.convs[op["conv"]]
    H, W, C, _ = g.tensors[op["inp"]]
    if (c["k"], c["stride"], op["pad"]) != (3, 1, 1) or c["cin"] != c["cout"] or c["cout"] > 64 or c["cin"] % 8 or W % 2:
        return False
    if op.get("group") is not None or op.get("residual") or op.get("out_dtype", "bf16") != "bf16" or C != c["cin"]:
        return False
    p = _C.attach_splitk_workspace(_C.ConvProblem(), splitk_ws)
    p.opts = opts
    p.R = p.S = 3
    p.stride_h = p.stride_w = p.pad_top = p.pad_left = 1
    p.act, p.out_dtype, p.num_segments = _C.RN_ACT_NONE, _C.RN_DT_BF16, 1
    s = p.seg[0]
    s.N, s.H, s.W, s.Cin, s.pix_stride, s.Ho, s.Wo, s.Cout = B, H, W // 2, 2 * C, 2 * C, H, W // 2, 2 * C
    return lib.rn_conv_kernel_id(ctypes.byref(p)) == 3


def stem_pool_partner(g, stem_op, stem_k):
    """The MaxPool op that rn_stem_conv_bn_relu_pool can absorb: the ResNet stem (7x7/2, 64 channels, relu | relu6)
    whose only consumer is a 3x3 / stride-2 pool with SAME pads (resnet.py:288-307); None otherwise (EfficientNet's
    3x3 swish stem has no pool).  RNET_FUSE_STEM_POOL=0 keeps the two launches."""
    if os.environ.get("RNET_FUSE_STEM_POOL", "1") == "0":
        return None
    c = g.convs[stem_op["conv"]]
    if stem_k != 7 or c["cout"] != 64 or stem_op.get("act") not in ("relu", "relu6"):
        return None
    users = [o for o in g.ops if stem_op["out"] in
             [v for k, vv in o.items() if k not in ("op", "out", "outs", "conv", "bn", "act")
              for v in (vv if isinstance(vv, (list, tuple)) else [vv]) if isinstance(v, str)]]
    if len(users) != 1 or users[0]["op"] != "maxpool":
        return None
    pool = users[0]
    if pool["k"] != 3 or pool["stride"] != 2 or pool["pad_top"] not in (0, 1) or pool["pad_left"] not in (0, 1):
        return None
    return pool


class InferenceEngine:
    def __init__(self, graph, variables, batch_size, device, bn_epsilon=1e-3, capture_graph=False, f16=False,
                 launch_opts=None):
        self.g = graph
        self.B = int(batch_size)
        self.dev = torch.device(device)
        self.eps = float(bn_epsilon)
        self.f16 = bool(f16)   # `mixed_float16`: IEEE-half activations on librnet_hip_f16.so, else bfloat16
        self.h16 = torch.float16 if self.f16 else torch.bfloat16
        self._DT = {"bf16": self.h16, "f32": torch.float32}
        self.lib = _C.lib(self.f16)
        # rn_launch_opts of THIS engine's conv launches (kernel-family overrides for tests / A/B timing; the library has
        # no process-wide knobs)
        if isinstance(launch_opts, dict):
            launch_opts = _C.LaunchOpts(**launch_opts)
        self.launch_opts = launch_opts.copy() if launch_opts is not None else _C.LaunchOpts()
        self._keep = []     # ctypes structs / arrays that must outlive the launches
        self.steps = []     # list of (callable, name)
        self.t = {}         # tensor name -> torch tensor
        self.packed = {}    # conv name -> packed bf16 weight
        self._pair = {}     # f32 conv name -> its weight planes are stacked along Cout (rn_conv_segment.w_pair)
        self._graph = None
        self._capture = bool(capture_graph)
        with torch.cuda.device(self.dev):
            self._alloc()
            # ResNet stage-1 bottleneck blocks as ONE launch each (rn_bottleneck64_fwd, retinanet/model/bottleneck.py)
            from .bottleneck import Bottleneck64, find_blocks
            self.bneck, self._bneck_skip = {}, set()      # first op's output name -> fused block; outputs of fused ops
            for blk in find_blocks(self.g):
                fb = Bottleneck64(self.lib, self.g, blk, self.B, self.dev, self.h16, self.launch_opts, self.t[blk["x"]],
                                  self.t[blk["name"]])
                if fb.ok:
                    self.bneck[blk["ops"][0]["out"]] = fb
                    self._bneck_skip.update(o["out"] for o in blk["ops"])
            # split-K of the persistent conv kernels' last round: the launches of this engine run in order on one stream
            self.splitk_ws = _C.new_splitk_workspace(self.lib, self.dev, default_on=True)
            self.load_variables(variables)
            # second stream (see _side_launch): its conv launches need a split-K workspace of their own — launches
            # that share one must be ordered on one stream
            # Measured (round 5, one box, tools/bench_infer.py): batch 8 3.554 -> 3.512 ms; batch 1 1.572 -> 1.594 ms — at
            # batch 1 a fork / join pair costs more than the ~15 us launch it takes off the chain, so the default is two
            # streams from batch 4 up.  RNET_INFER_STREAMS=1 / 2 forces either.
            ns = os.environ.get("RNET_INFER_STREAMS", "")
            self.two_streams = (ns == "2") if ns in ("1", "2") else self.B >= 4
            self.splitk_ws_side = (_C.new_splitk_workspace(self.lib, self.dev, default_on=True)
                                   if self.two_streams else None)
            self.side_steps = set()   # names of the launches that go to the second stream
            self.step_io = {}         # step name -> (tensor names read, tensor names written)
            self._build()
            self._side_stream = (_C.concurrent_stream(self.lib, self.dev, [torch.cuda.current_stream(self.dev)])[0]
                                 if self.side_steps else None)
            self._events = [torch.cuda.Event() for _ in range(2 * len(self.side_steps))]

    # ---- buffers ---------------------------------------------------------------------------
    def _alloc(self):
        for name, (H, W, C, dt) in self.g.tensors.items():
            self.t[name] = torch.empty((self.B, H, W, C), dtype=self._DT[dt], device=self.dev)
        # first-layer conv: the image is repacked to a zero-bordered bf16 NHWC4 buffer (rn_pack_image_nhwc4)
        stem = next(o for o in self.g.ops if o["op"] == "stem")
        Hs, Ws = self.g.tensors[stem["out"]][:2]
        k = stem.get("k", 7)
        self.stem_k, self.stem_pad = k, (stem.get("pad_top", 3), stem.get("pad_left", 3))
        H, W, _, _ = self.g.tensors["images"]
        self.Hp = max((Hs - 1) * 2 + k, H + self.stem_pad[0])
        self.Wp = -(-max((Ws - 1) * 2 + 8, W + self.stem_pad[1]) // 8) * 8
        self.stem_in = torch.empty((self.B, self.Hp, self.Wp, 4), dtype=self.h16, device=self.dev)
        se_ops = [o for o in self.g.ops if o["op"] == "se"]
        if se_ops:
            nbytes = max(self.lib.rn_se_workspace_bytes(self.B, self.g.ses[o["se"]]["C"]) for o in se_ops)
            self.se_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)

    # ---- weights -----------------------------------------------------------------------------
    def load_variables(self, variables):
        """(Re)pack conv kernels and refold BN/bias from a name -> tensor dict."""
        lib = self.lib
        st = _C.current_stream()
        self.fold = getattr(self, "fold", {})

        def stable(key, tensor):   # keep addresses stable for a captured graph
            old = self.packed.get(key)
            if old is None:
                self.packed[key] = tensor.contiguous()
            else:
                old.copy_(tensor)

        for op in self.g.ops:
            if op["op"] == "se":
                name = op["se"]
                se = self.g.ses[name]
                f32 = lambda n: variables[name + n].to(self.dev, torch.float32)
                stable(name + ":w1", f32("/conv2d/kernel").reshape(se["C"], se["se"]).t().to(self.h16))
                stable(name + ":b1", f32("/conv2d/bias"))
                stable(name + ":w2", f32("/conv2d_1/kernel").reshape(se["se"], se["C"]).t().to(self.h16))
                stable(name + ":b2", f32("/conv2d_1/bias"))
                continue
            if op["op"] == "dwconv":
                d = self.g.dws[op["dw"]]
                w = variables[d["kvar"]].to(self.dev, torch.float32).contiguous()
                buf = self.packed.get(op["dw"])
                if buf is None:
                    buf = torch.empty((d["k"] * d["k"], d["C"]), dtype=self.h16, device=self.dev)
                _C.check(lib.rn_pack_depthwise_weight(_C.ptr(w), d["k"], d["C"], _C.ptr(buf), st),
                         "rn_pack_depthwise_weight")
                self.packed[op["dw"]] = buf
                self._fold(op["out"], variables, op.get("bn"), None)
                continue
            if op["op"] not in ("conv", "stem"):
                continue
            if op["out"] in self._bneck_skip:      # part of a fused bottleneck block: packed below
                continue
            cname = op["conv"]
            c = self.g.convs[cname]
            w = variables[c.get("kvar", cname + "/kernel")].to(self.dev, torch.float32).contiguous()
            cout_pad = lib.rn_conv_cout_pad(c["cout"])
            if self._pixel_pair(op):    # 64-channel 3x3 layer as a 128 -> 128 convolution over pixel pairs
                w2 = pixel_pair_kernel(w).contiguous()
                buf = self.packed.get(cname)
                if buf is None:
                    buf = torch.empty((lib.rn_conv_cout_pad(2 * c["cout"]), 3, 3, lib.rn_conv_cin_pad(2 * c["cin"])),
                                      dtype=self.h16, device=self.dev)
                _C.check(lib.rn_pack_conv_weight(_C.ptr(w2), 3, 3, 2 * c["cin"], 2 * c["cout"],
                                                 lib.rn_conv_cin_pad(2 * c["cin"]), _C.ptr(buf), st), "rn_pack_conv_weight")
                self.packed[cname] = buf
                self._fold(op["out"], variables, op.get("bn"), variables.get(cname + "/bias"), repeat=2)
                continue
            if cname not in self.packed or True:
                if op["op"] == "stem":
                    buf = self.packed.get(cname)
                    k = self.stem_k
                    if buf is None:
                        buf = torch.empty((cout_pad, k, 32), dtype=self.h16, device=self.dev)
                    _C.check(lib.rn_pack_stem_weight_rs(_C.ptr(w), k, k, c["cout"], _C.ptr(buf), st),
                             "rn_pack_stem_weight_rs")
                else:
                    buf = self.packed.get(cname)
                    cin_pad = lib.rn_conv_cin_pad(c["cin"])
                    terms = self._w_terms(op)
                    if self._w_pair(op):   # narrow f32 layer (box prediction): the two planes along Cout
                        if buf is None:
                            buf = torch.empty((lib.rn_conv_pair_rows(c["cout"]), c["k"], c["k"], cin_pad), dtype=self.h16,
                                              device=self.dev)
                        _C.check(lib.rn_pack_conv_weight_pair(_C.ptr(w), 0, c["k"], c["k"], c["cin"], c["cout"], cin_pad,
                                                              _C.ptr(buf), st), "rn_pack_conv_weight_pair")
                        self.packed[cname] = buf
                        self._fold(op["out"], variables, op.get("bn"), variables.get(cname + "/bias"))
                        continue
                    if buf is None:
                        buf = torch.empty((cout_pad, c["k"], c["k"], terms * cin_pad), dtype=self.h16,
                                          device=self.dev)
                    if terms > 1:   # f32 layer (detection_head.py:80-88): its f32 kernel as split-bf16 planes
                        _C.check(lib.rn_pack_conv_weight_split(_C.ptr(w), 0, c["k"], c["k"], c["cin"], c["cout"], cin_pad,
                                                               terms, _C.ptr(buf), st), "rn_pack_conv_weight_split")
                    else:
                        _C.check(lib.rn_pack_conv_weight(_C.ptr(w), c["k"], c["k"], c["cin"], c["cout"], cin_pad,
                                                         _C.ptr(buf), st), "rn_pack_conv_weight")
                self.packed[cname] = buf
            self._fold(op["out"], variables, op.get("bn"), variables.get(cname + "/bias"))
        for fb in self.bneck.values():
            fb.load(variables, self.eps)

    def _pixel_pair(self, op):
        """True when conv `op` runs in pixel-pair form (pixel_pair_ok): decided once per op"""
        self._ppair = getattr(self, "_ppair", {})
        key = op["out"]
        if key not in self._ppair:
            self._ppair[key] = pixel_pair_ok(self.lib, self.g, op, self.B, self.launch_opts,
                                             getattr(self, "splitk_ws", None))
        return self._ppair[key]

    def _w_terms(self, op):
        """split-bf16 weight planes of the dtype=float32 prediction convs (rn_conv_segment.w_terms); 1 elsewhere"""
        if self._w_pair(op):
            return 1
        return _C.PRED_W_TERMS if op.get("out_dtype") == "f32" and op["op"] == "conv" else 1

    def _w_pair(self, op):
        """True for a narrow f32 conv whose two weight planes go along Cout (rn_conv_segment.w_pair; the reasoning is in
        train_engine.TrainEngine._pair_form): decided once per conv from the shapes of the grouped launch it runs in."""
        if op.get("out_dtype") != "f32" or op["op"] != "conv":
            return False
        cname = op["conv"]
        if cname not in self._pair:
            c = self.g.convs[cname]
            ops = [o for o in self.g.ops if o["op"] == "conv" and o["conv"] == cname]
            groups = {o.get("group") for o in ops}
            ok = self.lib.rn_conv_cout_pad(c["cout"]) <= 64 and len(groups) == 1 and None not in groups
            if ok:
                tn = self.g.tensors
                shapes = [tn[o["inp"]][:2] + (tn[o["inp"]][2],) + tn[o["out"]][:2] for o in ops]
                ok = _C.pair_form_kernel(self.lib, self.B, c["k"], c["stride"], ops[0]["pad"], c["cin"], c["cout"], shapes,
                                         self.launch_opts) > 0
            self._pair[cname] = ok
        return self._pair[cname]

    def _fold(self, key, variables, bn, bias, repeat=1):
        """(scale, shift, bias) of the conv epilogue: the Conv2D layer's bias stays separate (it is added before the
        layer's output is rounded to bf16), BN inference = x*scale + shift with scale = gamma/sqrt(var+eps),
        shift = beta - mean*scale."""
        bias = None if bias is None else bias.to(self.dev, torch.float32)
        scale = shift = None
        if bn:
            gamma = variables[bn + "/gamma"].to(self.dev, torch.float32)
            beta = variables[bn + "/beta"].to(self.dev, torch.float32)
            mean = variables[bn + "/moving_mean"].to(self.dev, torch.float32)
            var = variables[bn + "/moving_variance"].to(self.dev, torch.float32)
            scale = gamma / torch.sqrt(var + self.eps)
            shift = beta - mean * scale
        if repeat > 1:     # pixel-pair form: the per-channel vectors once per pixel of the pair
            scale, shift, bias = [None if t is None else t.repeat(repeat) for t in (scale, shift, bias)]
        new = (scale, shift, bias)
        old = self.fold.get(key)
        if old is None:
            self.fold[key] = [None if t is None else t.contiguous() for t in new]
        else:  # keep addresses stable for a captured graph
            for dst, src in zip(old, new):
                if src is not None:
                    dst.copy_(src)

    # ---- launch list -------------------------------------------------------------------------
    def _conv_segment(self, seg, op):
        c = self.g.convs[op["conv"]]
        x, y = self.t[op["inp"]], self.t[op["out"]]
        scale, shift, bias = self.fold[op["out"]]
        seg.x = x.data_ptr()
        seg.w = self.packed[op["conv"]].data_ptr()
        seg.y = y.data_ptr()
        seg.scale = scale.data_ptr() if scale is not None else None
        seg.shift = shift.data_ptr() if shift is not None else None
        seg.bias = bias.data_ptr() if bias is not None else None
        seg.w_terms = self._w_terms(op)
        seg.w_pair = 1 if self._w_pair(op) else 0
        seg.residual = self.t[op["residual"]].data_ptr() if op.get("residual") else None
        seg.N, seg.H, seg.W, seg.Cin = self.B, x.shape[1], x.shape[2], c["cin"]
        seg.pix_stride = x.shape[3]
        seg.Ho, seg.Wo, seg.Cout = y.shape[1], y.shape[2], c["cout"]
        if self._pixel_pair(op):       # the same bytes as [N, H, W/2, 2C]
            seg.W, seg.Wo, seg.Cin, seg.Cout, seg.pix_stride = x.shape[2] // 2, y.shape[2] // 2, 2 * c["cin"], 2 * c["cout"], 2 * x.shape[3]

    def _tensor_users(self, tname):
        """(op, role) of every op of the graph that reads tensor `tname`; role = the op field that names it"""
        users = []
        for o in self.g.ops:
            for k, v in o.items():
                if k in ("op", "out", "outs", "conv", "bn", "act", "group", "dw", "se", "out_dtype"):
                    continue
                if o["op"] in ("balance", "se") and k in ("tensors", "tensor"):
                    k = "inplace"
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    if isinstance(t, str) and t == tname:
                        users.append((o, k))
        return users

    def _side_launch(self, ops, second_of_split):
        """Launches that leave the main stream.  At batch 1 (BASELINE configs[0]: the reference's latency protocol) every
        launch is a fraction of the chip wide and ~15 us of fixed latency long, so a launch that nothing on the critical
        path waits for can run BESIDE it on a second stream — in a captured graph a parallel branch:
          * the projection shortcut of a ResNet stage's first block (resnet.py:220-228): read again only as the `residual`
            of the block's last conv, three launches later;
          * a prediction conv whose outputs are network outputs, next to the other head's (detection_head.py:80-88);
          * the second launch of a grouped launch that split_by_depth cut in two.
        Results are what they were: every launch computes the same tiles from the same inputs.  RNET_INFER_STREAMS=1
        keeps one stream (A/B)."""
        if not self.two_streams:
            return False
        if second_of_split:
            return True
        outs_ = [o["out"] for o in ops]
        roles = [r for t in outs_ for _, r in self._tensor_users(t)]
        if roles and all(r == "residual" for r in roles):
            return True
        net_outs = {n for d in self.g.outputs.values() for n in d.values()}
        if not roles and all(t in net_outs for t in outs_):
            # only when another launch follows that does not need it (the class head's prediction conv)
            later = [o for o in self.g.ops if o["op"] == "conv" and o["out"] in net_outs and o["out"] not in outs_]
            idx = {id(o): i for i, o in enumerate(self.g.ops)}
            return any(idx[id(o)] > max(idx[id(q)] for q in ops) for o in later)
        return False

    def _add_conv_launch(self, ops, second_of_split=False):
        first = ops[0]
        c0 = self.g.convs[first["conv"]]
        side = self._side_launch(ops, second_of_split)
        p = _C.attach_splitk_workspace(_C.ConvProblem(), self.splitk_ws_side if side else self.splitk_ws)
        p.opts = self.launch_opts
        p.R = p.S = c0["k"]
        p.stride_h = p.stride_w = c0["stride"]
        p.pad_top = p.pad_left = first["pad"]
        p.act = _C.ACT_IDS[first["act"]]
        p.out_dtype = _C.RN_DT_F32 if first["out_dtype"] == "f32" else _C.RN_DT_BF16
        p.num_segments = len(ops)
        for i, op in enumerate(ops):
            c = self.g.convs[op["conv"]]
            if (c["k"], c["stride"], op["pad"], op["act"], op["out_dtype"]) != \
                    (c0["k"], c0["stride"], first["pad"], first["act"], first["out_dtype"]):
                raise ValueError(f"conv group {first.get('group')} mixes shapes")
            self._conv_segment(p.seg[i], op)
        self._keep.append(p)
        lib = self.lib
        pref = ctypes.byref(p)
        name = first.get("group") or first["out"]
        self.conv_problems = getattr(self, "conv_problems", {})
        if "conv:" + name in self.conv_problems:      # second launch of a split group (split_by_depth)
            name += ":rest"
        self.conv_problems["conv:" + name] = p   # for profilers: lib.rn_conv_tile_rows(byref(p))

        def run(st):
            _C.check(lib.rn_conv2d_nhwc_fwd(pref, st), f"rn_conv2d_nhwc_fwd[{name}]")
        self.steps.append((run, "conv:" + name))
        self.step_io["conv:" + name] = ({t for o in ops for t in (o["inp"], o.get("residual")) if t}, {o["out"] for o in ops})
        if side:
            self.side_steps.add("conv:" + name)

    def _add_dw_launch(self, ops):
        first = ops[0]
        d0 = self.g.dws[first["dw"]]
        p = _C.DwProblem()
        p.k, p.stride, p.pad_top, p.pad_left = d0["k"], d0["stride"], first["pad_top"], first["pad_left"]
        p.act = _C.ACT_IDS[first["act"]]
        p.num_segments = len(ops)
        for i, op in enumerate(ops):
            d = self.g.dws[op["dw"]]
            if (d["k"], d["stride"], op["pad_top"], op["pad_left"], op["act"]) != \
                    (d0["k"], d0["stride"], first["pad_top"], first["pad_left"], first["act"]):
                raise ValueError(f"depthwise group {first.get('group')} mixes shapes")
            x, y = self.t[op["inp"]], self.t[op["out"]]
            scale, shift, _ = self.fold[op["out"]]
            s = p.seg[i]
            s.x, s.w, s.y = x.data_ptr(), self.packed[op["dw"]].data_ptr(), y.data_ptr()
            s.scale = scale.data_ptr() if scale is not None else None
            s.shift = shift.data_ptr() if shift is not None else None
            s.N, s.H, s.W, s.C, s.Ho, s.Wo = self.B, x.shape[1], x.shape[2], d["C"], y.shape[1], y.shape[2]
        self._keep.append(p)
        lib = self.lib
        pref = ctypes.byref(p)
        name = first.get("group") or first["out"]

        def run(st):
            _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(pref, st), f"rn_depthwise_conv2d_nhwc_fwd[{name}]")
        self.steps.append((run, "dwconv:" + name))
        self.step_io["dwconv:" + name] = ({o["inp"] for o in ops}, {o["out"] for o in ops})

    def _build(self):
        lib = self.lib
        B = self.B
        done_groups = set()
        fused_pools = set()   # MaxPool outputs written by the fused stem launch
        for op in self.g.ops:
            kind = op["op"]
            if kind == "stem":
                img = self.t["images"]
                H, W = img.shape[1], img.shape[2]
                y = self.t[op["out"]]
                c = self.g.convs[op["conv"]]
                pin, pout, pimg = self.stem_in.data_ptr(), y.data_ptr(), img.data_ptr()

                def pack(st, pimg=pimg, pin=pin, H=H, W=W, pt=self.stem_pad[0], pl=self.stem_pad[1]):
                    _C.check(lib.rn_pack_image_nhwc4(pimg, B, H, W, pt, pl, self.Hp, self.Wp, pin, st),
                             "rn_pack_image_nhwc4")
                self.steps.append((pack, "pack_stem_input"))
                self.step_io["pack_stem_input"] = ({"images"}, {":stem_in"})
                p = _C.attach_splitk_workspace(_C.ConvProblem(), self.splitk_ws)
                p.opts = self.launch_opts
                p.R, p.S, p.stride_h, p.stride_w, p.pad_top, p.pad_left = self.stem_k, 1, 2, 2, 0, 0
                p.act = _C.ACT_IDS[op["act"]]
                p.out_dtype = _C.RN_DT_BF16
                p.num_segments = 1
                s = p.seg[0]
                scale, shift, _ = self.fold[op["out"]]
                s.x, s.w, s.y = pin, self.packed[op["conv"]].data_ptr(), pout
                s.scale, s.shift, s.residual = scale.data_ptr(), shift.data_ptr(), None
                s.N, s.H, s.W, s.Cin, s.pix_stride = B, self.Hp, self.Wp, 32, 4
                s.Ho, s.Wo, s.Cout = y.shape[1], y.shape[2], c["cout"]
                self._keep.append(p)
                pref = ctypes.byref(p)
                pool = stem_pool_partner(self.g, op, self.stem_k)
                if pool is not None:   # ResNet: stem + BatchNorm + relu + MaxPool in one launch, the stem output stays on chip
                    z = self.t[pool["out"]]
                    fa = (pin, s.w, s.scale, s.shift, z.data_ptr(), B, self.Hp, self.Wp, y.shape[1], y.shape[2], self.stem_k,
                          c["cout"], p.act, pool["k"], pool["stride"], pool["pad_top"], pool["pad_left"], z.shape[1], z.shape[2])
                    fused_pools.add(pool["out"])

                    def stem_pool(st, fa=fa):
                        _C.check(lib.rn_stem_conv_bn_relu_pool(*fa, st), "rn_stem_conv_bn_relu_pool")
                    self.steps.append((stem_pool, "conv:stem"))
                    self.step_io["conv:stem"] = ({":stem_in"}, {pool["out"]})
                    continue

                def stem(st, pref=pref):
                    _C.check(lib.rn_conv2d_nhwc_fwd(pref, st), "rn_conv2d_nhwc_fwd[stem]")
                self.steps.append((stem, "conv:stem"))
                self.step_io["conv:stem"] = ({":stem_in"}, {op["out"]})
            elif kind == "conv" and op["out"] in self._bneck_skip:
                fb = self.bneck.get(op["out"])
                if fb is not None:                 # the block's first op in graph order carries the launch
                    self.steps.append((fb.launch, fb.name))
                    self.step_io[fb.name] = ({fb.blk["x"]}, {fb.blk["name"]})
            elif kind == "conv":
                grp = op.get("group")
                if grp is None:
                    self._add_conv_launch([op])
                elif grp not in done_groups:
                    done_groups.add(grp)
                    for j, sub in enumerate(split_by_depth(self.g, [o for o in self.g.ops if o["op"] == "conv" and o.get("group") == grp])):
                        self._add_conv_launch(sub, second_of_split=j > 0)
            elif kind == "dwconv":
                grp = op.get("group")
                if grp is None:
                    self._add_dw_launch([op])
                elif grp not in done_groups:
                    done_groups.add(grp)
                    self._add_dw_launch([o for o in self.g.ops if o["op"] == "dwconv" and o.get("group") == grp])
            elif kind == "se":
                x = self.t[op["tensor"]]
                name, se = op["se"], self.g.ses[op["se"]]
                args = (x.data_ptr(), B, x.shape[1] * x.shape[2], se["C"], self.packed[name + ":w1"].data_ptr(),
                        self.packed[name + ":b1"].data_ptr(), self.packed[name + ":w2"].data_ptr(),
                        self.packed[name + ":b2"].data_ptr(), se["se"], self.se_ws.data_ptr(), self.se_ws.numel())

                def se_run(st, args=args, name=name):
                    _C.check(lib.rn_squeeze_excite_inplace(*args, st), f"rn_squeeze_excite_inplace[{name}]")
                self.steps.append((se_run, "se:" + op["tensor"]))
                self.step_io["se:" + op["tensor"]] = ({op["tensor"]}, {op["tensor"]})
            elif kind == "maxpool":
                if op["out"] in fused_pools:
                    continue
                x, y = self.t[op["inp"]], self.t[op["out"]]
                args = (x.data_ptr(), y.data_ptr(), B, x.shape[1], x.shape[2], x.shape[3], op["k"], op["stride"],
                        op["pad_top"], op["pad_left"], y.shape[1], y.shape[2])

                def pool(st, args=args):
                    _C.check(lib.rn_maxpool2d_nhwc(*args, st), "rn_maxpool2d_nhwc")
                self.steps.append((pool, "maxpool:" + op["out"]))
                self.step_io["maxpool:" + op["out"]] = ({op["inp"]}, {op["out"]})
            elif kind == "topdown":
                ins = [self.t[n] for n in op["ins"]]
                outs = [self.t[n] for n in op["outs"]]
                pin, pout = _C.ptr_array(ins), _C.ptr_array(outs)
                self._keep += [pin, pout]
                H0, W0, C = ins[0].shape[1], ins[0].shape[2], ins[0].shape[3]
                act = _C.ACT_IDS[op["act"]]

                def td(st, pin=pin, pout=pout, L=len(ins), H0=H0, W0=W0, C=C, act=act):
                    _C.check(lib.rn_fpn_topdown(pin, pout, L, B, H0, W0, C, act, st), "rn_fpn_topdown")
                self.steps.append((td, "fpn_topdown"))
                self.step_io["fpn_topdown"] = (set(op["ins"]), set(op["outs"]))
            elif kind == "balance":
                ts = [self.t[n] for n in op["tensors"]]
                pin = _C.ptr_array(ts)
                self._keep.append(pin)
                mid = op["mid"]
                scratch = torch.empty_like(ts[mid])
                self._keep.append(scratch)
                H0, W0, C = ts[0].shape[1], ts[0].shape[2], ts[0].shape[3]

                def bal(st, pin=pin, L=len(ts), mid=mid, H0=H0, W0=W0, C=C, sp=scratch.data_ptr()):
                    _C.check(lib.rn_balance_features(pin, pin, L, mid, B, H0, W0, C, sp, st),
                             "rn_balance_features")
                self.steps.append((bal, "balance_features"))
                self.step_io["balance_features"] = (set(op["tensors"]), set(op["tensors"]))
            else:
                raise ValueError(kind)
        self.outputs = {k: {lv: self.t[n] for lv, n in d.items()} for k, d in self.g.outputs.items()}

    # ---- run -----------------------------------------------------------------------------------
    def _launch_all(self):
        st = _C.current_stream()
        if not self.side_steps:
            for fn, _ in self.steps:
                fn(st)
            return
        # two streams (see _side_launch): a side launch is ordered behind everything the main stream has enqueued so far
        # (event fork), the main stream waits for it in front of the first launch that touches what it wrote (or
        # overwrites what it read), and at the end.  Under torch.cuda.graph the events become graph edges.
        main, side = torch.cuda.current_stream(), self._side_stream
        sst = ctypes.c_void_p(side.cuda_stream)
        pending, ne = [], 0
        for fn, name in self.steps:
            reads, writes = self.step_io.get(name, (None, None))
            if name in self.side_steps:
                fork, done = self._events[ne], self._events[ne + 1]
                ne += 2
                fork.record(main)
                side.wait_event(fork)
                fn(sst)
                done.record(side)
                pending.append((done, reads, writes))
                continue
            keep = []
            for done, r_s, w_s in pending:
                if reads is None or (w_s & (reads | writes)) or (r_s & writes):
                    main.wait_event(done)
                else:
                    keep.append((done, r_s, w_s))
            pending = keep
            fn(st)
        for done, _, _ in pending:
            main.wait_event(done)

    def __call__(self, images):
        """images f32[B,H,W,3] (already normalised) -> prediction dict of static output buffers."""
        if tuple(images.shape) != tuple(self.t["images"].shape):
            raise ValueError(f"expected images of shape {tuple(self.t['images'].shape)}, got {tuple(images.shape)}")
        with torch.cuda.device(self.dev):
            if images.data_ptr() != self.t["images"].data_ptr():
                self.t["images"].copy_(images, non_blocking=True)
            if self._capture:
                if self._graph is None:
                    self._launch_all()  # warm-up outside capture (lazy module loads, attribute sets)
                    torch.cuda.synchronize()
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._launch_all()
                self._graph.replay()
            else:
                self._launch_all()
        return self.outputs
