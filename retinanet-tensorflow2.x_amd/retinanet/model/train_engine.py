"""TrainEngine — the MI355X replacement of Executor._train_step (retinanet/executor.py:409-441).

One call = forward (training-mode BatchNorm) -> RetinaNetLoss forward+backward -> backward
through heads / BalanceFeatures / FPN / ResNet -> weight decay + per-tensor and global clipping
-> data-parallel all-reduce (RCCL) -> SGD momentum + EMA, all as HIP launches on static buffers.

There is no autograd tape: the backward pass is the closed-form gradient of every layer,
scheduled from the same static graph as the forward pass.
  * conv backward = wgrad (transpose-read MFMA kernel, deterministic split-K) + dgrad (the
    forward implicit-GEMM kernel run on dy with flipped/transposed weights; stride-2 layers go
    through a zero-insertion upsample), gradients of multi-consumer tensors are accumulated in
    the dgrad epilogue (residual add in place);
  * BatchNorm forward/backward are two-stage reductions; with `use_sync` and >1 rank the
    per-group [sum, sumsq] / [sum g, sum g*xhat] vectors are all-reduced between the stages
    (SyncBatchNormalization, model/utils.py:10-12);
  * frozen layers (training.freeze_variables, executor.py:154-176) keep inference-mode BN folded
    into their conv epilogue and get no backward at all;
  * parameters, gradients, momentum and EMA live in four flat fp32 arenas (conv kernels in the
    compute layout [Cout][R][S][Cin]); the optimizer is three multi-tensor launches and also
    refreshes the bf16 compute copy of every kernel.
The stem trains too (configs without `freeze_variables`): its weight gradient runs the same wgrad
kernel on the packed NHWC4 image (7 row taps x 32 = 8 column taps x 4 channels), the 3x3/2 SAME
max-pool has a gather-form backward.
"""
from __future__ import annotations

import ctypes
import math
import os

import numpy as np
import torch

from retinanet import _C
from .engine import split_by_depth

_DT = {"bf16": torch.bfloat16, "f32": torch.float32}
_SEG_DTYPE = np.dtype([("offset", "<i8"), ("size", "<i8"), ("wd", "<i4"), ("bb", "<i4"), ("nb", "<i4"),
                       ("pad", "<i4"), ("bf", "<i8")])


def _hwio_to_ohwi(w):
    return w.permute(3, 0, 1, 2).contiguous()


def _ohwi_to_hwio(w):
    return w.permute(1, 2, 3, 0).contiguous()


class TrainEngine:
    def __init__(self, model, batch_size, frozen_regexes=(), process_group=None, world_size=None, frozen_names=(),
                 launch_opts=None, wide_pred_terms=None, force_dp=None):
        """launch_opts: `_C.LaunchOpts` (or a dict of its fields) copied into every rn_conv_problem / rn_wgrad_problem of
        THIS engine (include/rnet_hip.h rn_launch_opts) — kernel-family overrides for tests and A/B timing; the library
        has no process-wide knobs.
        wide_pred_terms: bf16 weight planes the TRAINING forward of the wide dtype=float32 prediction conv (the class
        head's 720-channel layer, detection_head.py:80-88) multiplies by — 2 = the split-bf16 form of its f32 kernel
        (inference / export always use it), 1 = rb(w) only.  Default: RNET_TRAIN_PRED_W_TERMS, else
        params.training.prediction_weight_planes, else 2 (the reference's f32 layer).  History: round 5 made ONE plane the
        default on an A/B taken at the reference's initialisation, where the logits are bias + a small kernel term (spread
        0.33): class-loss moved by 3.1e-7 relative.  Round 6 repeated the A/B in the trained-detector regime — the class-
        prediction kernel scaled by 7, logit spread 2.3 (tools/ab_pred_planes.py --pred-scale 7,
        profiles/r06_ab_pred_planes_scale7.json): the class-loss moves by 4.3e-5 relative, above the 1e-5 the contract
        allows for the loss (gradient cosines stay > 0.99999) — so the default went back to two planes; one plane remains
        a documented opt-in worth 0.8 ms of the 29 ms step.  The narrow box-regression layer always keeps both planes.
        force_dp (default RNET_FORCE_DP=1): run the data-parallel machinery — SyncBN messages, the bucketed gradient
        all-reduce overlapped with the backward pass, the clip flag read — although this engine has ONE replica, over
        whatever process group it was given (a 1-rank `nccl` group: bench.py's `extra.dp_overhead`, the cost of that
        machinery on one GPU; losses and gradients equal the plain step's)."""
        self.model = model
        self.g = model.graph
        self.params_cfg = model.params
        self.B = int(batch_size)
        self.dev = model.device
        # `mixed_float16` (BASELINE config 5): IEEE-half activations / packed weights on librnet_hip_f16.so + the
        # LossScaleOptimizer arithmetic of optimizer_step; RNET_F16=0 keeps bfloat16 storage under that policy
        self.f16 = (str(getattr(getattr(model.params, "floatx", None), "precision", "")) == "mixed_float16"
                    and os.environ.get("RNET_F16", "1") != "0")
        self.h16 = torch.float16 if self.f16 else torch.bfloat16
        self._DT = {"bf16": self.h16, "f32": torch.float32}
        self.lib = _C.lib(self.f16)
        self.pg = process_group
        if world_size is None:   # one source of truth with RetinaNetLoss, which asks torch.distributed
            import torch.distributed as dist
            world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.world = int(world_size)
        bn = self.params_cfg.architecture.batch_norm
        self.eps, self.momentum_bn = float(bn.epsilon), float(bn.momentum)
        if force_dp is None:
            force_dp = os.environ.get("RNET_FORCE_DP", "0") == "1"
        self.dp_active = self.world > 1 or bool(force_dp)    # collectives are issued (over 1 rank when forced)
        self.sync_bn = bool(bn.use_sync) and self.dp_active
        if isinstance(launch_opts, dict):
            launch_opts = _C.LaunchOpts(**launch_opts)
        self.launch_opts = launch_opts.copy() if launch_opts is not None else _C.LaunchOpts()
        # data parallel: the persistent kernels leave a few CUs to RCCL (rn_launch_opts.reserved_cus)
        if self.dp_active and not self.launch_opts.reserved_cus:
            self.launch_opts.reserved_cus = int(os.environ.get("RNET_COMM_CUS", "8"))
        # the per-device handle (rn_create): device, CU count, this engine's launch defaults; owns the native communicators
        self.handle = _C.Handle(self.lib, self.dev.index if self.dev.index is not None else torch.cuda.current_device(),
                                self.launch_opts)
        if wide_pred_terms is None:
            wide_pred_terms = os.environ.get("RNET_TRAIN_PRED_W_TERMS") or \
                getattr(getattr(model.params, "training", None), "prediction_weight_planes", None) or 2
        self.wide_pred_terms = max(1, min(int(wide_pred_terms), _C.PRED_W_TERMS))
        self._pair_cache = {}
        self.frozen = set(frozen_names)
        for k in model.variables:
            if any(rx.search(k) for rx in frozen_regexes):
                self.frozen.add(k)
        self._keep = []
        self._train_step_active = False
        self._c2_local, self._c2_sent, self.c2_normalizer = None, False, None
        self._overlap_on = False
        self._overlap_done = 0
        self._step_args = dict(wdc=0.0, alpha=0.0, unscale=1.0, clip=0.0)
        self.native_comm = None   # retinanet.comm.NativeComm for the small per-layer messages (SyncBN, normaliser)
        self.native_comm_buckets = None   # ... and a second one for the gradient buckets (rn_allreduce_bucket), or None
        self._small_msgs = 0      # C3 messages (SyncBN sums; C2 rides in the first) sent since the step began
        self.syncbn_messages_per_step = None   # their count in the last train_step (bench.py: config.syncbn_messages)
        self._algo = {}           # id(rn_conv_problem) -> (algorithmic FLOPs, algorithmic bytes) where the launch executes more
        self.hbm_profile = None   # bench.py: list that collects (event0, event1, kernel name, algorithmic bytes)
        self.conv_launches = []   # (name, rn_conv_problem) of every implicit-GEMM launch: lib.rn_conv_kernel_id(byref(p))
        self.wgrad_launches = []  # (name, rn_wgrad_problem) of every weight-gradient launch
        cus = os.environ.get("RNET_WGRAD_CUS", "160,208")   # round 6, same-box A/B (profiles/r06_ab/wgrad_cus.txt): 176,256 28.23, 160,208 28.01, 144,192 28.02, 112,192 30.3 ms
        self._wgrad_cap = tuple(int(v) for v in cus.split(",")) if cus not in ("0", "") else None
        if os.environ.get("RNET_WGRAD_STREAM", "1") == "0":
            self._wgrad_cap = None      # one-stream backward: nothing to leave CUs to
        if self._wgrad_cap and len(self._wgrad_cap) == 1:
            self._wgrad_cap = (self._wgrad_cap[0], 2 * self._wgrad_cap[0])
        self._wgrad_capped = []   # (problem, capped target): set_wgrad_cap(False) lifts the cap (bench.py's exclusive step)
        self.step_count = 0
        self.conv_profile = None
        self.wgrad_profile = None   # bench.py: list that collects (event0, event1, algorithmic FLOPs, kernel) per wgrad launch
        # bench.py `roofline.layers`: list that collects (event0, event1, launch name, algorithmic FLOPs, algorithmic bytes,
        # kernel) for EVERY implicit-GEMM launch of the step — forward, data gradient, weight gradient
        self.layer_profile = None
        self.fuse_bn_stats = os.environ.get("RNET_FUSE_BN_STATS", "1") != "0"   # conv epilogue writes BN partial sums
        # data-gradient epilogue writes stage 1 of the BatchNorm backward reduction of the layer it produces dz for
        self.fuse_bn_bwd = os.environ.get("RNET_FUSE_BN_BWD", "1") != "0"
        # =2: also the multi-segment groups (the four head-tower depths).  Measured same-box: 3-4 more 122 us reduction
        # launches go, the eight 550 us tower data gradients get ~15 us longer each in the step: +0.25 % on the step
        # (round 3, with the weight-gradient CU cap: 31.06 -> 30.92 ms, +0.45 %, tools/probes/ab_env.sh) for -0.026 on the
        # dominant kernel's MFMA fraction, the figure bench.py's roofline reports — off by default, the capability stays tested
        self.fuse_bn_bwd_groups = os.environ.get("RNET_FUSE_BN_BWD", "1") == "2"
        self.bn_act_mask = os.environ.get("RNET_BN_ACT_MASK", "1") != "0"   # relu gates of the residual layers as bit masks
        self.bn_bwd_ws = {}       # id(rn_bn_problem) -> workspace that holds the externally written backward partials
        self._bn_bwd_pending = {}  # id(rn_bn_problem) -> segments whose dz-writing launch is planned (see _plan_dgrad_launch)
        self.bn_bwd_fused = []    # tensor names whose BatchNorm backward reduction runs in a dgrad epilogue
        # weight / bias gradient launches on a second HIP stream: nothing in the backward pass reads them, so they
        # run beside the data-gradient chain (MFMA-bound wgrad next to the HBM-bound BatchNorm backward kernels)
        self.side_stream_on = os.environ.get("RNET_WGRAD_STREAM", "1") != "0"
        self._side_stream = None
        self.drop_connect = True     # stochastic depth of the EfficientNet skip blocks (efficientnet.py:97-113)
        self.dc_masks = {}           # project conv output -> (f32[B] factors, survival_prob)
        self.dc_all = self.dc_p = None
        self.dc_generator = torch.Generator(device=self.dev)
        self.dc_generator.manual_seed(1337)
        self._prepare_graph()
        with torch.cuda.device(self.dev):
            # split-K of the persistent conv kernels' last round (rn_conv_problem.splitk_ws): every forward / data-gradient
            # launch runs on the main stream, in order, so one workspace serves them all; attached at creation because the
            # dispatcher looks at it
            self.splitk_ws = _C.new_splitk_workspace(self.lib, self.dev)
            self._analyse()
            self._alloc_params()
            self._alloc_tensors()
            self._fold_frozen()
            self._build_forward()
            self._build_backward()
            self._group_wgrad_steps()
            for name, cp in self.conv_launches:   # a launch writes stage-1 BatchNorm partials for all its segments or none
                marks = {bool(cp.seg[i].bn_bwd_y) for i in range(cp.num_segments)}
                if name.startswith("dgrad:") and len(marks) != 1:
                    raise RuntimeError(f"{name}: BatchNorm backward fusion covers only part of the launch's segments")

    # ------------------------------------------------------------------------------------------
    def _prepare_graph(self):
        """Engine-local copy of the op list: squeeze-excite runs out of place in training (its backward
        needs the un-gated input), so `se` ops get an output tensor `<t>:se` and later readers of `<t>` are
        redirected to it.  Also classifies every variable (compute layout, weight decay)."""
        g = self.g
        self.tensors = dict(g.tensors)
        self.ops = []
        alias = {}
        for op in g.ops:
            o = dict(op)
            for key in ("inp", "residual"):
                if o.get(key) in alias:
                    o[key] = alias[o[key]]
            if o["op"] == "se":
                t = o["tensor"]
                o["inp"], o["out"] = alias.get(t, t), t + ":se"
                self.tensors[o["out"]] = g.tensors[t]
                alias[t] = o["out"]
            if o["op"] in ("topdown",):
                o["ins"] = [alias.get(n, n) for n in o["ins"]]
            self.ops.append(o)
        # variable name -> (kind, layer): conv / dw / se1 / se2 kernels are weight-decayed
        # (executor.py:308-327: every trainable variable with 'kernel' in its name)
        self.var_kind = {}
        for cname, c in g.convs.items():
            self.var_kind[c.get("kvar", cname + "/kernel")] = ("conv", cname)
        for dname, d in getattr(g, "dws", {}).items():
            self.var_kind[d["kvar"]] = ("dw", dname)
        for sname in getattr(g, "ses", {}):
            self.var_kind[sname + "/conv2d/kernel"] = ("se1", sname)
            self.var_kind[sname + "/conv2d_1/kernel"] = ("se2", sname)

    def _kvar(self, op):
        if op["op"] == "dwconv":
            return self.g.dws[op["dw"]]["kvar"]
        c = self.g.convs[op["conv"]]
        return c.get("kvar", op["conv"] + "/kernel")

    def _conv_trainable(self, op):
        return self._kvar(op) not in self.frozen

    def _bn_trainable(self, op):
        return op.get("bn") and (op["bn"] + "/gamma") not in self.frozen

    def _analyse(self):
        self.requires = {"images": False}
        for op in self.ops:
            kind = op["op"]
            if kind in ("conv", "stem"):
                tr = self._conv_trainable(op) or bool(self._bn_trainable(op))
                ins = [op["inp"]] + ([op["residual"]] if op.get("residual") else [])
                self.requires[op["out"]] = tr or any(self.requires[i] for i in ins)
            elif kind == "maxpool":
                self.requires[op["out"]] = self.requires[op["inp"]]
            elif kind == "topdown":
                r = any(self.requires[i] for i in op["ins"])
                for o in op["outs"]:
                    self.requires[o] = r or self.requires.get(o, False)
            elif kind == "balance":
                pass
            elif kind == "dwconv":
                if not self._conv_trainable(op):
                    raise NotImplementedError("frozen depthwise layers are not a shipped configuration")
                self.requires[op["out"]] = True
            elif kind == "se":
                if any((op["se"] + sfx) in self.frozen for sfx in ("/conv2d/kernel", "/conv2d_1/kernel")):
                    raise NotImplementedError("frozen squeeze-excite layers are not a shipped configuration")
                self.requires[op["out"]] = True
            else:
                raise NotImplementedError(f"training through '{kind}' ops is not built")
        # a conv layer is "live" when its kernel trains; mixed frozen conv / live BN is not a shipped case
        for op in self.ops:
            if op["op"] in ("conv", "stem", "dwconv") and op.get("bn") and \
                    self._conv_trainable(op) != bool(self._bn_trainable(op)):
                raise NotImplementedError(f"{self._kvar(op)}: conv and its BatchNorm must be frozen together")

    # ---- flat parameter arenas ---------------------------------------------------------------------
    def _alloc_params(self):
        lib = self.lib
        chunk = lib.rn_optim_chunk()
        v = self.model.variables
        names = [k for k in v if self.g.var_specs[k].get("trainable", True) and k not in self.frozen]
        self.train_names = names
        segs, block_seg = [], []
        # the first 4 floats of every arena are reserved: G[0] / G[1] are the "a clip factor != 1 on some rank" /
        # "gradients not finite on some rank" slots that ride in the LAST gradient bucket's all-reduce (the bucket
        # at the front of the arena completes last in the backward pass)
        off, bf_off, nblk = 4, 0, 0
        self.p_off, self.bf_off = {}, {}     # bf_off: conv name | "dw:<name>" | "<se>:w1" / "<se>:w2" -> offset in Pbf
        self.fwd_packs = []                  # live convs whose Cin is not its own K-step padding: repacked per step
        self.fwd_pack_of = {}
        self.split_packs = []                # live f32 convs (prediction layers): split-bf16 planes, repacked per step
        self.split_pack_of = {}
        self.pair_packs = set()              # ... those of them whose two planes are stacked along Cout (w_pair)
        f32_convs = {o["conv"] for o in self.ops if o["op"] == "conv" and o.get("out_dtype") == "f32"}
        for i, k in enumerate(names):
            n = v[k].numel()
            kind, layer = self.var_kind.get(k, ("other", None))
            bfo = -1
            if kind == "conv":
                c = self.g.convs[layer]
                if c["cin"] == 3:
                    pass                                        # first-layer conv: its own packed form
                elif layer in f32_convs and self._f32_terms(layer) > 1:
                    cinp = lib.rn_conv_cin_pad(c["cin"])        # detection_head.py:80-88: the layer keeps its f32 kernel
                    if self._pair_form(layer):                  # narrow layer (box prediction): the planes along Cout
                        buf = torch.zeros((lib.rn_conv_pair_rows(c["cout"]), c["k"], c["k"], cinp), dtype=self.h16,
                                          device=self.dev)
                        self.pair_packs.add(layer)
                    else:
                        buf = torch.zeros((lib.rn_conv_cout_pad(c["cout"]), c["k"], c["k"], self._f32_terms(layer) * cinp),
                                          dtype=self.h16, device=self.dev)
                    self.split_packs.append((k, c, cinp, buf))
                    self.split_pack_of[layer] = buf
                elif lib.rn_conv_cin_pad(c["cin"]) == c["cin"]:
                    bfo = bf_off                                # plain cast of the master = the compute layout
                    self.bf_off[layer] = bf_off
                    bf_off += lib.rn_conv_cout_pad(c["cout"]) * c["k"] * c["k"] * c["cin"]
                else:
                    cinp = lib.rn_conv_cin_pad(c["cin"])
                    buf = torch.zeros((lib.rn_conv_cout_pad(c["cout"]), c["k"], c["k"], cinp), dtype=self.h16,
                                      device=self.dev)
                    self.fwd_packs.append((k, c, cinp, buf))
                    self.fwd_pack_of[layer] = buf
            elif kind == "dw":
                bfo = bf_off
                self.bf_off["dw:" + layer] = bf_off
                bf_off += n
            elif kind in ("se1", "se2"):
                bfo = bf_off
                self.bf_off[layer + (":w1" if kind == "se1" else ":w2")] = bf_off
                bf_off += n
            bf_off = (bf_off + 7) // 8 * 8
            nb = (n + chunk - 1) // chunk
            segs.append((off, n, 1 if kind != "other" else 0, nblk, nb, 0, bfo))   # executor.py:308-327: kernels only
            block_seg += [i] * nb
            self.p_off[k] = (off, n)
            off += (n + 3) // 4 * 4
            nblk += nb
        self.n_params = off
        self.P = torch.zeros((off,), dtype=torch.float32, device=self.dev)
        self.G = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.E = torch.zeros_like(self.P)
        self.Pbf = torch.zeros((max(bf_off, 8),), dtype=self.h16, device=self.dev)
        self._bf_copies = [(self.p_off[k], s[6]) for k, s in zip(names, segs) if s[6] >= 0]
        seg_np = np.zeros((len(segs),), dtype=_SEG_DTYPE)
        for i, s in enumerate(segs):
            seg_np[i] = s
        self.segs_dev = torch.from_numpy(seg_np.view(np.uint8).copy()).to(self.dev)
        self.block_seg_dev = torch.tensor(block_seg, dtype=torch.int32, device=self.dev)
        self.n_blocks, self.n_segs = len(block_seg), len(segs)
        self.opt_ws = torch.empty((lib.rn_optim_workspace_bytes(self.n_blocks, self.n_segs),), dtype=torch.uint8,
                                  device=self.dev)
        self.metrics = torch.zeros((8,), dtype=torch.float32, device=self.dev)
        self._block_elems = [(segs[si][0] + (bi - segs[si][3]) * chunk, min(chunk, segs[si][1] - (bi - segs[si][3]) * chunk))
                             for bi, si in enumerate(block_seg)]   # (arena offset, elements) of every optimizer block
        self._seg_blocks = {k: (s[3], s[4]) for k, s in zip(names, segs)}   # variable -> (first block, blocks)
        self.loss_scale = None     # LossScaleOptimizer state (mixed_float16 configs): see optimizer_step
        self.load_from_model()

    def _pview(self, name, arena=None):
        off, n = self.p_off[name]
        return (self.P if arena is None else arena)[off:off + n]

    def _to_compute_layout(self, k, t):
        kind = self.var_kind.get(k, ("other", None))[0]
        return _hwio_to_ohwi(t) if kind in ("conv", "se1", "se2") else t

    def load_from_model(self):
        """model.variables (Keras layouts) -> flat arenas (conv / SE kernels as [Cout][R][S][Cin], depthwise
        kernels as [k*k][C]) + bf16 compute copies."""
        v = self.model.variables
        for k in self.train_names:
            t = self._to_compute_layout(k, v[k].to(self.dev, torch.float32))
            self._pview(k).copy_(t.reshape(-1))
        self.E.copy_(self.P)
        self.V.zero_()
        for (off, n), bfo in self._bf_copies:
            self.Pbf[bfo:bfo + n].copy_(self.P[off:off + n])
        self.refresh_packs()

    def _stem_op(self):
        return next(o for o in self.ops if o["op"] == "stem")

    def refresh_packs(self):
        """per-step repacks from the f32 masters: the live first-layer conv ([Cout][R][S][3] ->
        bf16 [Cout_pad][R rows][8 taps x 4 ch]) and live convs whose Cin is zero-padded to the K step."""
        st = _C.current_stream()
        op = self._stem_op()
        if self._conv_trainable(op):
            c = self.g.convs[op["conv"]]
            k = c["k"]
            if getattr(self, "stem_packed", None) is None:
                self.stem_packed = torch.zeros((self.lib.rn_conv_cout_pad(c["cout"]), k, 32), dtype=self.h16,
                                               device=self.dev)
            w = _ohwi_to_hwio(self._pview(self._kvar(op)).reshape(c["cout"], k, k, 3))
            _C.check(self.lib.rn_pack_stem_weight_rs(_C.ptr(w), k, k, c["cout"], _C.ptr(self.stem_packed), st),
                     "rn_pack_stem_weight_rs")
        for (kname, c, cinp, buf) in self.fwd_packs:
            off, _ = self.p_off[kname]
            _C.check(self.lib.rn_pack_conv_weight_ohwi(self.P.data_ptr() + 4 * off, c["k"], c["k"], c["cin"], c["cout"],
                                                       cinp, buf.data_ptr(), st), "rn_pack_conv_weight_ohwi")
        for (kname, c, cinp, buf) in self.split_packs:
            off, _ = self.p_off[kname]
            if self.var_kind[kname][1] in self.pair_packs:
                _C.check(self.lib.rn_pack_conv_weight_pair(self.P.data_ptr() + 4 * off, 1, c["k"], c["k"], c["cin"],
                                                           c["cout"], cinp, buf.data_ptr(), st), "rn_pack_conv_weight_pair")
                continue
            _C.check(self.lib.rn_pack_conv_weight_split(self.P.data_ptr() + 4 * off, 1, c["k"], c["k"], c["cin"],
                                                        c["cout"], cinp, self._f32_terms(self.var_kind[kname][1]),
                                                        buf.data_ptr(), st), "rn_pack_conv_weight_split")

    def refresh_stem_pack(self):
        self.refresh_packs()

    def _keras_layout(self, k, arena):
        """Variable `k` of a flat arena in the layout Keras holds it (conv / SE kernels HWIO)."""
        v = self.model.variables[k]
        t = self._pview(k, arena)
        if self.var_kind.get(k, ("other", None))[0] in ("conv", "se1", "se2"):
            kh, kw, ci, co = v.shape
            t = _ohwi_to_hwio(t.reshape(co, kh, kw, ci))
        return t.reshape(v.shape)

    def optimizer_slots(self):
        """{(variable, slot): f32 array}: the SGD `momentum` accumulators and the moving-average `average` copies
        (optimizers/builder.py:27-71 — what `save_weights` stores next to the weights for a resume)."""
        out = {}
        for k in self.train_names:
            out[(k, "momentum")] = self._keras_layout(k, self.V).detach().cpu().numpy().copy()
            out[(k, "average")] = self._keras_layout(k, self.E).detach().cpu().numpy().copy()
        return out

    def load_optimizer_slots(self, slots):
        """Inverse of `optimizer_slots`; variables without a stored slot keep their current state."""
        for (k, slot), arr in slots.items():
            if k not in self.p_off or slot not in ("momentum", "average"):
                continue
            t = self._to_compute_layout(k, torch.as_tensor(np.asarray(arr), dtype=torch.float32).to(self.dev))
            self._pview(k, self.V if slot == "momentum" else self.E).copy_(t.reshape(-1))

    def save_checkpoint(self, prefix):
        """Weights + BN moving statistics + optimizer slots (`momentum`, `average`) + the step counter, in
        TensorFlow's checkpoint FILE format (where executor.py:652-654 / 695-697 call `model.save_weights`).  Interop
        is one-way: this build reads what the reference wrote (by variable name through the object graph); the
        reference's Keras `load_weights` matches structurally and will not consume the flat object graph written
        here (retinanet/tf_checkpoint.py::save_weights)."""
        with torch.cuda.device(self.dev):
            self.store_to_model(use_ema=False)
            torch.cuda.synchronize()
        self.finish_step()
        extra = {"SGD/iter": np.asarray(self.step_count, dtype=np.int64)}
        if self.loss_scale:   # Keras' LossScaleOptimizer checkpoints its dynamic state the same way
            extra["loss_scale/current_loss_scale"] = np.asarray(self.loss_scale["scale"], dtype=np.float32)
            extra["loss_scale/good_steps"] = np.asarray(self.loss_scale["good"], dtype=np.int64)
        self.model.save_weights(prefix, slots=self.optimizer_slots(), extra=extra)

    def restore_checkpoint(self, prefix):
        """executor.py:221-244: load the latest weights and continue from their step."""
        with torch.cuda.device(self.dev):
            slots = self.model.load_weights(prefix)
            self.load_from_model()
            for bn, d in self.bn_state.items():
                d["mm"].copy_(self.model.variables[bn + "/moving_mean"])
                d["mv"].copy_(self.model.variables[bn + "/moving_variance"])
            self._fold_frozen()
            self.load_optimizer_slots(slots)
            it = self.model.loaded_extras.get("SGD/iter")
            self._ls_pending = False
            self.step_count = int(it) if it is not None else 0
            if self.model.optimizer is not None:
                self.model.optimizer.iterations = self.step_count
            ls = self.model.loaded_extras.get("loss_scale/current_loss_scale")
            opt = self.model.optimizer
            if ls is not None and opt is not None and opt.dynamic_loss_scale:   # a resumed mixed_float16 run keeps its scale
                good = self.model.loaded_extras.get("loss_scale/good_steps")
                self.loss_scale = dict(scale=float(ls), good=int(good) if good is not None else 0,
                                       growth_steps=int(opt.loss_scale_growth_steps), skipped=False)

    def store_to_model(self, use_ema=False):
        """flat arenas -> model.variables (executor.assign_moving_averaged_weights when use_ema)."""
        v = self.model.variables
        src = self.E if use_ema else self.P
        for k in self.train_names:
            v[k].copy_(self._keras_layout(k, src))
        for bn, d in self.bn_state.items():
            v[bn + "/moving_mean"].copy_(d["mm"])
            v[bn + "/moving_variance"].copy_(d["mv"])
        self.model._refresh()

    # ---- activations / gradients ---------------------------------------------------------------------
    def _alloc_tensors(self):
        B, dev = self.B, self.dev
        self.t, self.raw, self.grad = {}, {}, {}
        for name, (H, W, C, dt) in self.tensors.items():
            self.t[name] = torch.empty((B, H, W, C), dtype=self._DT[dt], device=dev)
        # first-layer conv: zero-bordered bf16 NHWC4 copy of the image (rn_pack_image_nhwc4)
        stem = self._stem_op()
        Hs, Ws = self.tensors[stem["out"]][:2]
        self.stem_k = stem.get("k", 7)
        self.stem_pad = (stem.get("pad_top", 3), stem.get("pad_left", 3))
        H, W, _, _ = self.tensors["images"]
        self.Hp = max((Hs - 1) * 2 + self.stem_k, H + self.stem_pad[0])
        self.Wp = -(-max((Ws - 1) * 2 + 8, W + self.stem_pad[1]) // 8) * 8
        self.stem_in = torch.empty((B, self.Hp, self.Wp, 4), dtype=self.h16, device=dev)
        for op in self.ops:
            if op["op"] in ("conv", "stem", "dwconv") and self._bn_trainable(op):
                self.raw[op["out"]] = torch.empty_like(self.t[op["out"]])
        # squeeze-excite: saved forward state per op + one shared workspace
        self.se_state = {}
        se_bytes = 0
        for op in self.ops:
            if op["op"] == "se":
                nb = self.lib.rn_se_workspace_bytes(B, self.g.ses[op["se"]]["C"])
                self.se_state[op["out"]] = torch.empty((nb,), dtype=torch.uint8, device=dev)
                se_bytes = max(se_bytes, nb)
        self.se_ws = torch.empty((max(se_bytes, 16),), dtype=torch.uint8, device=dev)
        # balance features runs out of place in training (its backward needs the inputs)
        self.bal_out = {}
        for op in self.ops:
            if op["op"] == "balance":
                for n in op["tensors"]:
                    self.bal_out[n] = torch.empty_like(self.t[n])
        # gradient buffers (bf16) for every tensor that needs one
        need = set()
        for op in self.ops:
            if op["op"] in ("conv", "dwconv", "se") and self.requires.get(op["out"]):
                need.add(op["out"])
                for i in [op["inp"]] + ([op["residual"]] if op.get("residual") else []):
                    if self.requires.get(i):
                        need.add(i)
            elif op["op"] in ("maxpool",) and self.requires.get(op["out"]):
                need.add(op["out"])
                if self.requires.get(op["inp"]):
                    need.add(op["inp"])
            elif op["op"] == "topdown":
                for n in op["ins"] + op["outs"]:
                    if self.requires.get(n):
                        need.add(n)
        for n in need:
            H, W, C, _ = self.tensors[n]
            self.grad[n] = torch.zeros((B, H, W, C), dtype=self.h16, device=dev)
        for n, t in self.bal_out.items():
            self.grad["bal:" + n] = torch.zeros_like(t)
        self.bn_state = {}
        for op in self.ops:
            if op["op"] in ("conv", "stem", "dwconv") and self._bn_trainable(op):
                bn = op["bn"]
                self.bn_state[bn] = {"mm": self.model.variables[bn + "/moving_mean"].to(dev, torch.float32).clone(),
                                     "mv": self.model.variables[bn + "/moving_variance"].to(dev, torch.float32).clone()}

    def _fold_frozen(self):
        """inference-mode scale/shift + packed weights for frozen conv(+BN) layers."""
        lib, v = self.lib, self.model.variables
        st = _C.current_stream()
        old_fold, old_packed = getattr(self, "fold", None), getattr(self, "packed_frozen", None)
        self.fold, self.packed_frozen = {}, {}
        for op in self.ops:
            if op["op"] not in ("conv", "stem") or self._conv_trainable(op):
                continue
            cname = op["conv"]
            c = self.g.convs[cname]
            w = v[self._kvar(op)].to(self.dev, torch.float32).contiguous()
            cp = lib.rn_conv_cout_pad(c["cout"])
            if op["op"] == "stem":
                buf = torch.empty((cp, c["k"], 32), dtype=self.h16, device=self.dev)
                _C.check(lib.rn_pack_stem_weight_rs(_C.ptr(w), c["k"], c["k"], c["cout"], _C.ptr(buf), st),
                         "rn_pack_stem_weight_rs")
            elif self._pixel_pair(op):     # 64-channel 3x3 layer as a 128 -> 128 convolution over pixel pairs (engine.pixel_pair_ok)
                from .engine import pixel_pair_kernel
                w2 = pixel_pair_kernel(w).contiguous()
                cinp = lib.rn_conv_cin_pad(2 * c["cin"])
                buf = torch.empty((lib.rn_conv_cout_pad(2 * c["cout"]), 3, 3, cinp), dtype=self.h16, device=self.dev)
                _C.check(lib.rn_pack_conv_weight(_C.ptr(w2), 3, 3, 2 * c["cin"], 2 * c["cout"], cinp, _C.ptr(buf), st),
                         "rn_pack_conv_weight")
            else:
                cinp = lib.rn_conv_cin_pad(c["cin"])
                buf = torch.empty((cp, c["k"], c["k"], cinp), dtype=self.h16, device=self.dev)
                _C.check(lib.rn_pack_conv_weight(_C.ptr(w), c["k"], c["k"], c["cin"], c["cout"], cinp,
                                                 _C.ptr(buf), st), "rn_pack_conv_weight")
            self.packed_frozen[cname] = buf
            bias = v.get(cname + "/bias")
            scale = shift = None
            if op.get("bn"):
                bn = op["bn"]
                gmm, bta = v[bn + "/gamma"].to(self.dev).float(), v[bn + "/beta"].to(self.dev).float()
                mean, var = v[bn + "/moving_mean"].to(self.dev).float(), v[bn + "/moving_variance"].to(self.dev).float()
                scale = (gmm / torch.sqrt(var + self.eps)).contiguous()
                shift = (bta - mean * scale).contiguous()
            # the Conv2D bias stays separate: it is added before the layer's output is rounded (rn_conv_segment)
            bias = None if bias is None else bias.to(self.dev).float().contiguous()
            if op["op"] == "conv" and self._pixel_pair(op):     # per-channel vectors once per pixel of the pair
                scale, shift, bias = [None if t is None else t.repeat(2).contiguous() for t in (scale, shift, bias)]
            self.fold[op["out"]] = (scale, shift, bias)
        # frozen ResNet stage-1 bottleneck blocks as ONE launch each (retinanet/model/bottleneck.py): every layer of the block
        # frozen with its BatchNorm, nothing below it needs a gradient
        if getattr(self, "bneck", None) is None:
            from .bottleneck import Bottleneck64, find_blocks
            self.bneck, self._bneck_skip = {}, set()
            mine = {o["out"]: o for o in self.ops if "out" in o}
            for blk in find_blocks(self.g):
                ops_ = [mine.get(o["out"]) for o in blk["ops"]]
                if any(o is None or self._conv_trainable(o) or self._bn_trainable(o) or self.requires.get(o["out"])
                       for o in ops_) or self.requires.get(blk["x"]):
                    continue
                fb = Bottleneck64(lib, self.g, blk, self.B, self.dev, self.h16, self.launch_opts, self.t[blk["x"]],
                                  self.t[blk["name"]])
                if fb.ok:
                    self.bneck[blk["ops"][0]["out"]] = fb
                    self._bneck_skip.update(o["out"] for o in blk["ops"])
        for fb in self.bneck.values():
            fb.load(v, self.eps)
        if old_fold is not None:
            # a refold after a restore: the launch descriptors hold the first buffers' addresses -> copy in place
            for k, buf in self.packed_frozen.items():
                old_packed[k].copy_(buf)
            for k, new3 in self.fold.items():
                for dst, src in zip(old_fold[k], new3):
                    if dst is not None:
                        dst.copy_(src)
            self.fold, self.packed_frozen = old_fold, old_packed

    # ---- helpers to build launches ---------------------------------------------------------------------
    def _conv_meta(self, p):
        """(ALGORITHMIC FLOPs, algorithmic HBM bytes, kernel variant or None) of a launch — SURVEY 8(d): FLOPs =
        2 * Ho*Wo*k*k*Cin*Cout of the LAYER per image (a data-gradient launch counts its layer's forward MACs, not the
        zero-upsampled or channel-padded GEMM it executes; split-bf16 weight planes count once); bytes = every input
        pixel read once + every output written once (+ residual read) + weights.  Variants bench.py tracks: the
        256x256x32 kernels and the 128x128x64 kernel, bf16 output."""
        osz = 4 if p.out_dtype == _C.RN_DT_F32 else 2
        dom = True
        for i in range(p.num_segments):
            s = p.seg[i]
            dom = dom and s.Cout > 64 and s.Cin % 64 == 0 and p.out_dtype == _C.RN_DT_BF16
        if id(p) in self._algo:
            flops, byts = self._algo[id(p)]
        else:
            flops = byts = 0
            for i in range(p.num_segments):
                s = p.seg[i]
                flops += 2 * s.N * s.Ho * s.Wo * p.R * p.S * s.Cin * s.Cout
                byts += 2 * s.N * s.H * s.W * s.pix_stride + osz * s.N * s.Ho * s.Wo * s.Cout + 2 * p.R * p.S * s.Cin * s.Cout
                if s.residual:
                    byts += 2 * s.N * s.Ho * s.Wo * s.Cout
        # one name per device symbol, so that a bench line and a rocprof kernel-stats row can be matched
        variant = None
        kid = self.lib.rn_conv_kernel_id(ctypes.byref(p))
        has_res = any(p.seg[i].residual for i in range(p.num_segments))
        bnb = bool(p.seg[0].bn_bwd_y)   # the BN_BWD variants (stage 1 of a BatchNorm backward reduction in the epilogue)
        tmpl = (f"<{'true' if p.out_dtype == _C.RN_DT_F32 else 'false'}, {'true' if has_res else 'false'}, "
                f"{'true' if bnb else 'false'}>")
        # (device symbols: conv_halo_kernel<f32 out, residual, BN_BWD, SPLIT, waves along the pixels>)
        if kid == 2:
            variant = "conv_halo_kernel" + tmpl[:-1] + ", false, 2> (256x256x32, 3x3 halo patch)"
        elif kid == 3:     # the same kernel template with 4 x 2 waves: 512 x 128 tiles (64 < Cout <= 128)
            variant = "conv_halo_kernel" + tmpl[:-1] + ", false, 4> (512x128x32, 3x3 halo patch)"
        elif kid == 1:
            variant = "conv_big_kernel" + tmpl + " (256x256x32)"
        elif dom:
            variant = "conv_fwd_kernel<128,128,64,bf16>"
        return flops, byts, variant

    def _launch_conv(self, p, st, what):
        """All implicit-GEMM launches (forward and dgrad) go through here so bench.py can bracket the
        dominant kernel variant with HIP events on the launch stream."""
        prof, lprof = self.conv_profile, self.layer_profile
        if prof is not None or lprof is not None:
            flops, byts, variant = self._conv_meta(p)
            if variant or lprof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _C.check(self.lib.rn_conv2d_nhwc_fwd(ctypes.byref(p), st), what)
                e1.record()
                if prof is not None and variant:
                    prof.append((e0, e1, flops, byts, variant))
                if lprof is not None:
                    if getattr(self, "_launch_names", None) is None or len(self._launch_names) != len(self.conv_launches):
                        self._launch_names = {id(q): n for n, q in self.conv_launches}
                    kid = self.lib.rn_conv_kernel_id(ctypes.byref(p))
                    kname = variant or ("conv_fwd_kernel (128-row tiles)" if kid == 0 else f"kernel id {kid}")
                    lprof.append((e0, e1, self._launch_names.get(id(p), what), flops, byts, kname))
                return
        _C.check(self.lib.rn_conv2d_nhwc_fwd(ctypes.byref(p), st), what)

    def _weight_ptr(self, cname):
        if cname in self.packed_frozen:
            return self.packed_frozen[cname].data_ptr()
        if cname in self.fwd_pack_of:
            return self.fwd_pack_of[cname].data_ptr()
        if cname in self.split_pack_of:
            return self.split_pack_of[cname].data_ptr()
        return self.Pbf.data_ptr() + 2 * self.bf_off[cname]

    def _pixel_pair(self, op):
        """True when the FROZEN conv `op` (inference form) runs as a convolution over pixel pairs (engine.pixel_pair_ok)"""
        from .engine import pixel_pair_ok
        self._ppair = getattr(self, "_ppair", {})
        key = op["out"]
        if key not in self._ppair:
            # tied to the inference-form launch: a frozen kernel in front of a LIVE BatchNorm (custom freeze regexes) goes
            # through _conv_problem(raw_mode=True), which keeps the layer's own shape — its weights must then not be the
            # pair-packed [128,3,3,128] buffer
            ok = (op["op"] == "conv" and not self._conv_trainable(op) and not self._bn_trainable(op)
                  and not self.requires.get(op["inp"]))
            self._ppair[key] = ok and pixel_pair_ok(self.lib, self.g, op, self.B, self.launch_opts, getattr(self, "splitk_ws", None))
        return self._ppair[key]

    def _f32_terms(self, layer):
        """bf16 weight planes of the dtype=float32 conv `layer` in THIS engine's forward pass: the narrow (pair-form)
        layers always carry both planes, the wide one `wide_pred_terms`"""
        if _C.PRED_W_TERMS <= 1:
            return 1
        return _C.PRED_W_TERMS if self._pair_form(layer) else self.wide_pred_terms

    def _pair_form(self, layer):
        if layer not in self._pair_cache:
            self._pair_cache[layer] = self._pair_form_uncached(layer)
        return self._pair_cache[layer]

    def _pair_form_uncached(self, layer):
        """True when the f32 conv `layer` (one kernel shared by the pyramid levels of a grouped launch) is narrow enough that
        its two weight planes go along Cout (rn_conv_segment.w_pair): 36 box-regression channels fill 72 of the 128 columns
        of the halo kernel's 512 x 128 tiles; along Cin they were a 64-column tile of the 128-row kernel at 2 x the K depth."""
        c = self.g.convs[layer]
        if self.lib.rn_conv_cout_pad(c["cout"]) > 64:
            return False                                        # wide layers (class prediction) already run 256-row tiles
        ops = [o for o in self.ops if o["op"] == "conv" and o["conv"] == layer]
        groups = {o.get("group") for o in ops}
        if len(groups) != 1 or None in groups:
            return False
        shapes = [self.tensors[o["inp"]][:2] + (self.tensors[o["inp"]][2],) + self.tensors[o["out"]][:2] for o in ops]
        return _C.pair_form_kernel(self.lib, self.B, c["k"], c["stride"], ops[0]["pad"], c["cin"], c["cout"], shapes,
                                   self.launch_opts) > 0

    def _group_ops(self, kind, grp):
        """The ops of one grouped launch, in graph order.  (The one group that mixes K depths — the FPN lateral 1x1 convs with
        512 / 1024 / 2048 input channels — is balanced inside rn_conv2d_nhwc_fwd: tiles numbered deepest segment first and
        dealt to the workgroups round-robin; sorting the list here alone made that launch slower)"""
        return [o for o in self.ops if o["op"] == kind and o.get("group") == grp]

    def _conv_problem(self, ops, dst_of, raw_mode):
        """forward conv launch over `ops`; raw_mode: write pre-BN output (+bias) without activation."""
        first = ops[0]
        c0 = self.g.convs[first["conv"]]
        assert not (raw_mode and any(self._pixel_pair(o) for o in ops)), "pair-packed weights behind a raw (pre-BN) launch"
        p = _C.attach_splitk_workspace(_C.ConvProblem(), self.splitk_ws)
        p.opts = self.launch_opts
        p.R = p.S = c0["k"]
        p.stride_h = p.stride_w = c0["stride"]
        p.pad_top = p.pad_left = first["pad"]
        p.act = _C.RN_ACT_NONE if raw_mode else _C.ACT_IDS[first["act"]]
        p.out_dtype = _C.RN_DT_F32 if first["out_dtype"] == "f32" else _C.RN_DT_BF16
        p.num_segments = len(ops)
        for i, op in enumerate(ops):
            c = self.g.convs[op["conv"]]
            x, y = self._src(op["inp"]), dst_of(op)
            s = p.seg[i]
            s.x, s.w, s.y = x.data_ptr(), self._weight_ptr(op["conv"]), y.data_ptr()
            s.scale = s.shift = s.bias = s.residual = None
            if raw_mode or self._conv_trainable(op):   # raw pre-BN output, or a live conv without BN (prediction convs)
                s.bias = self._pview(op["conv"] + "/bias").data_ptr() if c["bias"] else None
            else:
                sc, sh, bs = self.fold[op["out"]]
                s.scale = sc.data_ptr() if sc is not None else None
                s.shift = sh.data_ptr() if sh is not None else None
                s.bias = bs.data_ptr() if bs is not None else None
            if not raw_mode:
                s.residual = self.t[op["residual"]].data_ptr() if op.get("residual") else None
            s.w_terms = self._f32_terms(op["conv"]) if op["conv"] in self.split_pack_of and op["conv"] not in self.pair_packs else 1
            s.w_pair = 1 if op["conv"] in self.pair_packs else 0
            s.N, s.H, s.W, s.Cin, s.pix_stride = self.B, x.shape[1], x.shape[2], c["cin"], x.shape[3]
            s.Ho, s.Wo, s.Cout = y.shape[1], y.shape[2], c["cout"]
            if not raw_mode and self._pixel_pair(op):      # the same bytes as [N, H, W/2, 2C]; algorithmic work: the layer's own
                self._algo[id(p)] = (2 * self.B * y.shape[1] * y.shape[2] * 9 * c["cin"] * c["cout"],
                                     2 * x.numel() + 2 * y.numel() + 2 * 9 * c["cin"] * c["cout"])
                s.W, s.Wo, s.Cin, s.Cout, s.pix_stride = x.shape[2] // 2, y.shape[2] // 2, 2 * c["cin"], 2 * c["cout"], 2 * x.shape[3]
        self._keep.append(p)
        name = "fwd:" + (first.get("group") or first["out"])
        if any(n == name for n, _ in self.conv_launches):   # second launch of a split group (engine.split_by_depth)
            name += ":rest"
        self.conv_launches.append((name, p))
        return p

    def _dw_problem(self, ops, dst_of):
        """forward depthwise launch over `ops` writing the raw (pre-BN) or final output; bf16 weights are
        the plain-cast copies of the [k*k][C] masters."""
        d0 = self.g.dws[ops[0]["dw"]]
        p = _C.DwProblem()
        p.k, p.stride, p.pad_top, p.pad_left = d0["k"], d0["stride"], ops[0]["pad_top"], ops[0]["pad_left"]
        p.act, p.num_segments = _C.RN_ACT_NONE, len(ops)
        for i, op in enumerate(ops):
            d = self.g.dws[op["dw"]]
            if (d["k"], d["stride"], op["pad_top"], op["pad_left"]) != (p.k, p.stride, p.pad_top, p.pad_left):
                raise ValueError("depthwise group mixes shapes")
            x, y = self._src(op["inp"]), dst_of(op)
            s = p.seg[i]
            s.x, s.w, s.y = x.data_ptr(), self.Pbf.data_ptr() + 2 * self.bf_off["dw:" + op["dw"]], y.data_ptr()
            s.scale, s.shift, s.residual = None, None, None
            s.N, s.H, s.W, s.C, s.Ho, s.Wo = self.B, x.shape[1], x.shape[2], d["C"], y.shape[1], y.shape[2]
        self._keep.append(p)
        return p

    def _allreduce_small(self, t):
        """SyncBatchNorm / normaliser message: rn_allreduce_small on the compute stream when a native communicator was
        handed in (retinanet.comm.maybe_enable_native), torch.distributed otherwise"""
        self._small_msgs += 1
        if self.native_comm is not None:
            self.native_comm.all_reduce_small(t)
        else:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.pg)

    def _bn_pass(self, kind, pb, fn):
        """BatchNorm elementwise / reduction passes (HBM-bound): bench.py brackets them with events on the launch stream.
        Algorithmic bytes: apply = read y + write z (+ residual); bwd_reduce = read y + dz (+ z for the residual
        layers' gate); bwd_apply = the same reads + write dy (+ dres, + its old value when accumulating)."""
        prof = self.hbm_profile
        if prof is None or (kind == "bn_bwd_reduce" and pb.seg[0].ext_chunks_bwd > 0):   # only the final pass is left
            return fn()
        byts = 0
        for i in range(pb.num_segments):
            s = pb.seg[i]
            n = int(s.P) * int(s.C) * 2
            gate = (n // 16 if s.act_mask else n) if (s.residual and pb.act) else 0   # z, or its one-bit gate
            if kind == "bn_apply":
                byts += n * (3 if s.residual else 2) + (n // 16 if s.act_mask else 0)
            elif kind == "bn_bwd_reduce":
                byts += n * 2 + gate
            else:
                byts += n * 2 + gate + n * (1 + ((2 if s.dres_accumulate else 1) if s.dres else 0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.dev))
        fn()
        e1.record(torch.cuda.current_stream(self.dev))
        prof.append((e0, e1, kind, byts))

    def _bn_stats_finalize(self, prb, ws, sums, st):
        """Batch statistics -> (mean, invstd, scale, shift) + moving statistics.  One replica: the final reduction
        kernel also finalizes; SyncBN: the [2][C] sums are all-reduced between the two."""
        lib = self.lib
        if not self.sync_bn and os.environ.get("RNET_FUSE_BN_FINALIZE", "1") != "0":
            _C.check(lib.rn_bn_stats_finalize(prb, _C.ptr(ws), ws.numel(), st), "rn_bn_stats_finalize")
            return
        _C.check(lib.rn_bn_stats(prb, _C.ptr(ws), ws.numel(), st), "rn_bn_stats")
        if self.sync_bn:
            from retinanet.distribute import syncbn_merge
            # C2 (the loss normaliser's scalar all-reduce, retinanet_loss.py:46-49) rides in the spare slot of the step's
            # FIRST SyncBN message instead of being a collective of its own
            fold = self._c2_local is not None and not self._c2_sent
            norm = syncbn_merge(sums, self.world, self._allreduce_small, self._c2_local if fold else None)
            if fold:
                self._c2_sent = True
                self.c2_normalizer = norm
        _C.check(lib.rn_bn_finalize(prb, st), "rn_bn_finalize")

    def _bn_problem(self, ops, conv_problem=None):
        """BatchNorm problem over `ops`.  With `conv_problem` (the launch that produces the raw outputs): the forward
        statistics' stage-1 partial sums are written by the conv epilogue (rn_conv_segment.bn_partial, one row per
        128 output pixels) and rn_bn_stats only does the final ordered reduction."""
        p = _C.BnProblem()
        p.num_segments = len(ops)
        p.act = _C.ACT_IDS[ops[0]["act"]]
        p.bessel = 0 if self.sync_bn else 1
        p.eps, p.momentum, p.count_scale = self.eps, self.momentum_bn, float(self.world if self.sync_bn else 1)
        csum = sum(self.tensors[o["out"]][2] for o in ops)
        sums = torch.zeros((2 * csum + 1,), dtype=torch.float32, device=self.dev)   # + the slot C2 rides in
        bsums = torch.zeros((2 * csum,), dtype=torch.float32, device=self.dev)
        fwd = torch.zeros((4 * csum,), dtype=torch.float32, device=self.dev)
        off = 0
        dys = []
        for i, op in enumerate(ops):
            C = self.tensors[op["out"]][2]
            bn = op["bn"]
            y, z = self.raw[op["out"]], self.t[op["out"]]
            dy = torch.empty_like(y)
            dys.append(dy)
            s = p.seg[i]
            s.y, s.z, s.dy = y.data_ptr(), z.data_ptr(), dy.data_ptr()
            s.residual = self.t[op["residual"]].data_ptr() if op.get("residual") else None
            s.dz = self.grad[op["out"]].data_ptr() if op["out"] in self.grad else None
            s.dres = None
            s.sums = sums.data_ptr() + 4 * 2 * off
            s.bsums = bsums.data_ptr() + 4 * 2 * off
            s.fwd = fwd.data_ptr() + 4 * 4 * off
            s.gamma = self._pview(bn + "/gamma").data_ptr()
            s.beta = self._pview(bn + "/beta").data_ptr()
            s.moving_mean = self.bn_state[bn]["mm"].data_ptr()
            s.moving_var = self.bn_state[bn]["mv"].data_ptr()
            s.dgamma = self._pview(bn + "/gamma", self.G).data_ptr()
            s.dbeta = self._pview(bn + "/beta", self.G).data_ptr()
            s.P, s.C, s.dres_accumulate = y.shape[0] * y.shape[1] * y.shape[2], C, 0
            if s.residual and op["act"] in ("relu", "relu6") and self.bn_act_mask:
                # relu behind the residual add: the forward stores the gate as one bit per element, the two backward
                # passes read P*C/8 bytes instead of z (rn_bn_segment.act_mask)
                mk = torch.empty((int(s.P) * C // 8,), dtype=torch.uint8, device=self.dev)
                self._keep.append(mk)
                s.act_mask = mk.data_ptr()
            if op.get("survival") is not None and self.drop_connect:
                if self.dc_all is None:     # one [blocks, B] tensor so that a step draws every factor at once
                    nsurv = sum(1 for o in self.ops if o.get("survival") is not None)
                    self.dc_all = torch.ones((nsurv, self.B), dtype=torch.float32, device=self.dev)
                    self.dc_p = torch.ones((nsurv, 1), dtype=torch.float32, device=self.dev)
                j = len(self.dc_masks)
                m = self.dc_all[j]
                self.dc_p[j, 0] = float(op["survival"])
                self.dc_masks[op["out"]] = (m, float(op["survival"]))
                s.sample_scale, s.rows_per_sample = m.data_ptr(), y.shape[1] * y.shape[2]
            off += C
        fused = conv_problem is not None and self.fuse_bn_stats and conv_problem.out_dtype == _C.RN_DT_BF16
        if fused:
            for i in range(len(ops)):   # one row of partial sums per 128 output pixels of the kernel the dispatcher will run
                p.seg[i].ext_chunks = self.lib.rn_conv_bn_row_blocks(ctypes.byref(conv_problem), i)
        # (zeros: the ticket counters in the workspace tail start at zero, rnet_hip.h: rn_bn_workspace_bytes)
        ws = torch.zeros((max(self.lib.rn_bn_workspace_bytes(ctypes.byref(p)), 256),), dtype=torch.uint8,
                         device=self.dev)
        if fused:
            for i in range(len(ops)):
                conv_problem.seg[i].bn_partial = ws.data_ptr() + self.lib.rn_bn_partial_offset_bytes(ctypes.byref(p), i)
        self._keep += [p, sums, bsums, fwd, ws] + dys
        return p, sums, bsums, ws, dys

    # ---- forward --------------------------------------------------------------------------------------
    def _build_forward(self):
        lib, B = self.lib, self.B
        self.fwd_steps = []
        self.fused_pools = set()   # MaxPool outputs written by rn_stem_conv_bn_relu_pool
        self.bn_groups = {}   # first op out -> (problem, sums, bsums, ws, dys, ops)
        self.bal_src = {}     # tensor name -> balance output tensor (consumers read the balanced copy)
        done = set()
        for op in self.ops:
            kind = op["op"]
            if kind == "stem":
                img, y = self.t["images"], self.t[op["out"]]
                H, W = img.shape[1], img.shape[2]
                c = self.g.convs[op["conv"]]
                pin = self.stem_in.data_ptr()
                self._images_ptr = img.data_ptr()   # forward() points this at the caller's batch when it can be read in place
                self.fwd_steps.append(lambda st, pin=pin, H=H, W=W: _C.check(
                    lib.rn_pack_image_nhwc4(self._images_ptr, B, H, W, self.stem_pad[0], self.stem_pad[1], self.Hp, self.Wp,
                                            pin, st), "rn_pack_image_nhwc4"))
                live = self._conv_trainable(op)
                p = _C.attach_splitk_workspace(_C.ConvProblem(), self.splitk_ws)
                p.opts = self.launch_opts
                p.R, p.S, p.stride_h, p.stride_w, p.pad_top, p.pad_left = self.stem_k, 1, 2, 2, 0, 0
                p.act = _C.RN_ACT_NONE if live else _C.ACT_IDS[op["act"]]
                p.out_dtype, p.num_segments = _C.RN_DT_BF16, 1
                s = p.seg[0]
                if live:
                    s.x, s.w, s.y = pin, self.stem_packed.data_ptr(), self.raw[op["out"]].data_ptr()
                    s.scale, s.shift, s.residual = None, None, None
                else:
                    sc, sh, _ = self.fold[op["out"]]
                    s.x, s.w, s.y = pin, self.packed_frozen[op["conv"]].data_ptr(), y.data_ptr()
                    s.scale, s.shift, s.residual = sc.data_ptr(), sh.data_ptr(), None
                s.N, s.H, s.W, s.Cin, s.pix_stride = B, self.Hp, self.Wp, 32, 4
                s.Ho, s.Wo, s.Cout = y.shape[1], y.shape[2], c["cout"]
                self._keep.append(p)
                if live:
                    pb, sums, bsums, ws, dys = self._bn_problem([op])
                    self.bn_groups[op["out"]] = (pb, sums, bsums, ws, dys, [op])
                    prb = ctypes.byref(pb)

                    def run_stem(st, p=p, prb=prb, ws=ws, sums=sums, pb=pb):
                        self._launch_conv(p, st, "stem(train)")
                        self._bn_stats_finalize(prb, ws, sums, st)
                        self._bn_pass("bn_apply", pb, lambda: _C.check(lib.rn_bn_apply(prb, st), "rn_bn_apply"))
                    self.fwd_steps.append(run_stem)
                else:
                    from .engine import stem_pool_partner
                    pool = stem_pool_partner(self.g, op, self.stem_k)
                    if pool is not None and not self.requires.get(op["out"]):
                        # frozen stem (`resnet_initial`): conv + folded BatchNorm + relu + MaxPool in one launch
                        z = self.t[pool["out"]]
                        fa = (pin, s.w, s.scale, s.shift, z.data_ptr(), B, self.Hp, self.Wp, y.shape[1], y.shape[2],
                              self.stem_k, c["cout"], p.act, pool["k"], pool["stride"], pool["pad_top"], pool["pad_left"],
                              z.shape[1], z.shape[2])
                        self.fused_pools.add(pool["out"])
                        self.fwd_steps.append(lambda st, fa=fa: _C.check(lib.rn_stem_conv_bn_relu_pool(*fa, st),
                                                                         "rn_stem_conv_bn_relu_pool"))
                    else:
                        self.fwd_steps.append(lambda st, p=p: self._launch_conv(p, st, "stem"))
            elif kind == "conv" and op["out"] in self._bneck_skip:
                fb = self.bneck.get(op["out"])
                if fb is not None:                 # the block's first op in graph order carries the launch
                    def run_block(st, fb=fb):
                        lprof = self.layer_profile
                        if lprof is None:
                            fb.launch(st)
                            return
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        fb.launch(st)
                        e1.record()
                        lprof.append((e0, e1, "fwd:" + fb.name, fb.flops, fb.bytes, "bneck64_kernel (one launch per block)"))
                    self.fwd_steps.append(run_block)
            elif kind == "conv":
                grp = op.get("group")
                if grp is not None:
                    if grp in done:
                        continue
                    done.add(grp)
                    ops = self._group_ops("conv", grp)
                else:
                    ops = [op]
                live_bn = bool(self._bn_trainable(ops[0]))
                if any(bool(self._bn_trainable(o)) != live_bn for o in ops):
                    raise NotImplementedError("a conv group mixes frozen and live BatchNorm")
                if live_bn:
                    pc = self._conv_problem(ops, lambda o: self.raw[o["out"]], raw_mode=True)
                    pb, sums, bsums, ws, dys = self._bn_problem(ops, conv_problem=pc)
                    self.bn_groups[ops[0]["out"]] = (pb, sums, bsums, ws, dys, ops)
                    prb = ctypes.byref(pb)

                    def run(st, pc=pc, prb=prb, ws=ws, sums=sums, pb=pb):
                        self._launch_conv(pc, st, "conv(train)")
                        self._bn_stats_finalize(prb, ws, sums, st)
                        self._bn_pass("bn_apply", pb, lambda: _C.check(lib.rn_bn_apply(prb, st), "rn_bn_apply"))
                    self.fwd_steps.append(run)
                else:
                    for sub in split_by_depth(self.g, ops):
                        pc = self._conv_problem(sub, lambda o: self.t[o["out"]], raw_mode=False)
                        self.fwd_steps.append(lambda st, pc=pc: self._launch_conv(pc, st, "conv"))
            elif kind == "dwconv":
                grp = op.get("group")
                if grp is not None:
                    if grp in done:
                        continue
                    done.add(grp)
                    ops = [o for o in self.ops if o["op"] == "dwconv" and o.get("group") == grp]
                else:
                    ops = [op]
                live_bn = bool(self._bn_trainable(ops[0]))
                pd = self._dw_problem(ops, (lambda o: self.raw[o["out"]]) if live_bn else (lambda o: self.t[o["out"]]))
                prd = ctypes.byref(pd)
                if live_bn:
                    pb, sums, bsums, ws, dys = self._bn_problem(ops)
                    self.bn_groups[ops[0]["out"]] = (pb, sums, bsums, ws, dys, ops)
                    prb = ctypes.byref(pb)

                    def run_dw(st, prd=prd, prb=prb, ws=ws, sums=sums, pb=pb):
                        _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(prd, st), "depthwise(train)")
                        self._bn_stats_finalize(prb, ws, sums, st)
                        self._bn_pass("bn_apply", pb, lambda: _C.check(lib.rn_bn_apply(prb, st), "rn_bn_apply"))
                    self.fwd_steps.append(run_dw)
                else:
                    if ops[0].get("act") not in (None, "none"):
                        raise NotImplementedError("depthwise conv with an activation but no BatchNorm")
                    self.fwd_steps.append(lambda st, prd=prd: _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(prd, st),
                                                                       "depthwise"))
            elif kind == "se":
                x, y = self.t[op["inp"]], self.t[op["out"]]
                name, se = op["se"], self.g.ses[op["se"]]
                state = self.se_state[op["out"]]
                a = (x.data_ptr(), y.data_ptr(), B, x.shape[1] * x.shape[2], se["C"],
                     self.Pbf.data_ptr() + 2 * self.bf_off[name + ":w1"], self._pview(name + "/conv2d/bias").data_ptr(),
                     self.Pbf.data_ptr() + 2 * self.bf_off[name + ":w2"], self._pview(name + "/conv2d_1/bias").data_ptr(),
                     se["se"], state.data_ptr(), state.numel())
                self.fwd_steps.append(lambda st, a=a: _C.check(lib.rn_squeeze_excite_fwd(*a, st), "rn_squeeze_excite_fwd"))
            elif kind == "maxpool":
                if op["out"] in self.fused_pools:   # written by the fused stem launch
                    continue
                x, y = self.t[op["inp"]], self.t[op["out"]]
                args = (x.data_ptr(), y.data_ptr(), B, x.shape[1], x.shape[2], x.shape[3], op["k"], op["stride"],
                        op["pad_top"], op["pad_left"], y.shape[1], y.shape[2])
                self.fwd_steps.append(lambda st, a=args: _C.check(lib.rn_maxpool2d_nhwc(*a, st), "maxpool"))
            elif kind == "topdown":
                ins, outs = [self.t[n] for n in op["ins"]], [self.t[n] for n in op["outs"]]
                pin, pout = _C.ptr_array(ins), _C.ptr_array(outs)
                self._keep += [pin, pout]
                a = (pin, pout, len(ins), B, ins[0].shape[1], ins[0].shape[2], ins[0].shape[3], _C.ACT_IDS[op["act"]])
                self.fwd_steps.append(lambda st, a=a: _C.check(lib.rn_fpn_topdown(*a, st), "rn_fpn_topdown"))
            elif kind == "balance":
                ts = [self.t[n] for n in op["tensors"]]
                outs = [self.bal_out[n] for n in op["tensors"]]
                for n in op["tensors"]:
                    self.bal_src[n] = self.bal_out[n]
                pin, pout = _C.ptr_array(ts), _C.ptr_array(outs)
                self.bal_avg = torch.empty_like(ts[op["mid"]])
                self._keep += [pin, pout]
                a = (pin, pout, len(ts), op["mid"], B, ts[0].shape[1], ts[0].shape[2], ts[0].shape[3],
                     self.bal_avg.data_ptr())
                self.fwd_steps.append(lambda st, a=a: _C.check(lib.rn_balance_features(*a, st), "rn_balance_features"))
        # the tower convs were built before `balance` registered its outputs: rebuild is avoided by
        # resolving inputs lazily — _conv_problem reads self.bal_src at build time, so build again
        # for the groups that consume balanced tensors
        if self.bal_src:
            self._rebuild_consumers_of_balanced()
        self.outputs = {k: {lv: self.t[n] for lv, n in d.items()} for k, d in self.g.outputs.items()}

    def _rebuild_consumers_of_balanced(self):
        # the graph lists `balance` before the heads, so bal_src was already populated when the tower
        # groups were built (ops are visited in order); nothing to do.  Kept as an assertion.
        order = [o["op"] for o in self.ops]
        bi = order.index("balance")
        for o in self.ops[:bi]:
            if o["op"] == "conv":
                assert o["inp"] not in self.bal_src

    def _src(self, name):
        return self.bal_src.get(name, self.t[name])

    def _gradbuf(self, name):
        return self.grad["bal:" + name] if name in self.bal_src else self.grad[name]

    # ---- backward ---------------------------------------------------------------------------------------
    def _build_backward(self):
        lib, B = self.lib, self.B
        self.bwd_steps = []
        self.grad_init = {}       # tensor -> runtime flag "gradient buffer already written this step"
        self.dgrad_packs = []     # (master offset, conv dims, packed buffer)
        ops = self.ops
        self.dw_flip_packs = []   # (master offset, k, C, packed bf16 tap-reversed filter)
        first_of_group, seen = {}, set()
        for i, op in enumerate(ops):
            if op["op"] in ("conv", "dwconv") and op.get("group") and (op["op"], op["group"]) not in seen:
                seen.add((op["op"], op["group"]))
                first_of_group[(op["op"], op["group"])] = i
        plan = []
        for i in range(len(ops) - 1, -1, -1):
            op = ops[i]
            if op["op"] in ("conv", "dwconv"):
                grp = op.get("group")
                if grp is None:
                    plan.append((op["op"], [op]))
                elif first_of_group[(op["op"], grp)] == i:
                    plan.append((op["op"], self._group_ops(op["op"], grp)))
            elif op["op"] in ("maxpool", "topdown", "balance", "stem", "se"):
                plan.append((op["op"], op))
        # which tensor gradients get more than one contribution is decided at build time
        written = set()

        def mark(name):
            first = name not in written
            written.add(name)
            return first

        for kind, item in plan:
            if kind == "conv":
                self._plan_conv_backward(item, mark)
            elif kind == "dwconv":
                self._plan_dw_backward(item, mark)
            elif kind == "se":
                op = item
                x, dy, dx = self.t[op["inp"]], self.grad[op["out"]], self.grad[op["inp"]]
                mark(op["inp"])
                name, se = op["se"], self.g.ses[op["se"]]
                a = (x.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, x.shape[1] * x.shape[2], se["C"],
                     self.Pbf.data_ptr() + 2 * self.bf_off[name + ":w1"],
                     self.Pbf.data_ptr() + 2 * self.bf_off[name + ":w2"], se["se"], self.se_state[op["out"]].data_ptr(),
                     self._pview(name + "/conv2d/kernel", self.G).data_ptr(),
                     self._pview(name + "/conv2d/bias", self.G).data_ptr(),
                     self._pview(name + "/conv2d_1/kernel", self.G).data_ptr(),
                     self._pview(name + "/conv2d_1/bias", self.G).data_ptr(), self.se_ws.data_ptr(), self.se_ws.numel())
                se_bwd = lambda st, a=a: _C.check(lib.rn_squeeze_excite_bwd(*a, st), "rn_squeeze_excite_bwd")
                se_bwd.writes = [name + sfx for sfx in ("/conv2d/kernel", "/conv2d/bias", "/conv2d_1/kernel", "/conv2d_1/bias")]
                self.bwd_steps.append(se_bwd)
            elif kind == "maxpool":
                op = item
                if not self.requires.get(op["inp"]):
                    continue
                x, dy, dx = self.t[op["inp"]], self.grad[op["out"]], self.grad[op["inp"]]
                acc = 0 if mark(op["inp"]) else 1
                a = (x.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, x.shape[1], x.shape[2], x.shape[3], op["k"],
                     op["stride"], op["pad_top"], op["pad_left"], dy.shape[1], dy.shape[2], acc)
                self.bwd_steps.append(lambda st, a=a: _C.check(lib.rn_maxpool2d_nhwc_bwd(*a, st), "maxpool_bwd"))
            elif kind == "stem":
                op = item
                if not self._conv_trainable(op):
                    continue
                pb, sums, bsums, ws, dys, _ = self.bn_groups[op["out"]]
                pb.seg[0].dz = self.grad[op["out"]].data_ptr()
                c = self.g.convs[op["conv"]]
                H, W = self.t["images"].shape[1], self.t["images"].shape[2]
                pw = _C.WgradProblem()
                pw.opts = self.launch_opts
                k = self.stem_k
                pw.R, pw.S, pw.stride_h, pw.stride_w, pw.pad_top, pw.pad_left, pw.num_segments = k, 1, 2, 2, 0, 0, 1
                sg = pw.seg[0]
                sg.x, sg.dy = self.stem_in.data_ptr(), dys[0].data_ptr()
                sg.N, sg.H, sg.W, sg.Cin, sg.Ho, sg.Wo, sg.Cout = B, self.Hp, self.Wp, 32, dys[0].shape[1], dys[0].shape[2], c["cout"]
                sg.x_pix_stride = 4
                wsw = torch.empty((max(lib.rn_wgrad_workspace_bytes(ctypes.byref(pw)), 256),), dtype=torch.uint8,
                                  device=self.dev)
                dwp = torch.zeros((c["cout"], k, 8, 4), dtype=torch.float32, device=self.dev)
                gview = self._pview(self._kvar(op), self.G).view(c["cout"], k, k, 3)
                self._keep += [pw, wsw, dwp]

                def stem_bwd(st, prb=ctypes.byref(pb), ws=ws, bsums=bsums, pw=pw, wsw=wsw, dwp=dwp, gview=gview, k=k):
                    _C.check(lib.rn_bn_bwd_reduce(prb, _C.ptr(ws), ws.numel(), st), "rn_bn_bwd_reduce")
                    if self.sync_bn:
                        self._allreduce_small(bsums)
                    _C.check(lib.rn_bn_bwd_apply(prb, st), "rn_bn_bwd_apply")
                    _C.check(lib.rn_conv2d_nhwc_wgrad(ctypes.byref(pw), dwp.data_ptr(), 0.0, wsw.data_ptr(),
                                                      wsw.numel(), st), "stem wgrad")
                    gview.copy_(dwp[:, :, :k, :3])   # [co][r][8 taps x 4 ch] -> [co][r][s][c]
                stem_bwd.writes = [self._kvar(op), op["bn"] + "/gamma", op["bn"] + "/beta"]
                self.bwd_steps.append(stem_bwd)
            elif kind == "topdown":
                op = item
                L = len(op["ins"])
                act = _C.ACT_IDS[op["act"]]
                prev = None
                for l in range(L):
                    dout = self.grad[op["outs"][l]]
                    din = self.grad[op["ins"][l]]
                    mark(op["ins"][l])
                    outp = self.t[op["outs"][l]].data_ptr() if l < L - 1 else None
                    a = (dout.data_ptr(), prev, outp, din.data_ptr(), B, dout.shape[1], dout.shape[2], dout.shape[3],
                         act if l < L - 1 else _C.RN_ACT_NONE)
                    self.bwd_steps.append(lambda st, a=a: _C.check(lib.rn_fpn_topdown_bwd_level(*a, st), "topdown_bwd"))
                    prev = din.data_ptr()
            elif kind == "balance":
                op = item
                names = op["tensors"]
                dout = [self.grad["bal:" + n] for n in names]
                ins = [self.t[n] for n in names]
                din = [self.grad[n] for n in names]
                for n in names:
                    mark(n)
                nsc = lib.rn_balance_features_bwd_scratch_bytes(len(names), op["mid"], B, ins[0].shape[1], ins[0].shape[2],
                                                                ins[0].shape[3])
                scratch = torch.empty((max(int(nsc), 256),), dtype=torch.uint8, device=self.dev)
                pd, pi, pn = _C.ptr_array(dout), _C.ptr_array(ins), _C.ptr_array(din)
                self._keep += [pd, pi, pn, scratch]
                a = (pd, pi, pn, self.bal_avg.data_ptr(), scratch.data_ptr(), scratch.numel(), len(names), op["mid"], B,
                     ins[0].shape[1], ins[0].shape[2], ins[0].shape[3])
                self.bwd_steps.append(lambda st, a=a: _C.check(lib.rn_balance_features_bwd(*a, st), "balance_bwd"))

    def set_wgrad_cap(self, on):
        """The CU cap of the weight-gradient launches (see _plan_conv_backward) on / off: with the chip to themselves
        (bench.py's one-stream `exclusive` step) they run uncapped; the workspaces fit either split-K plan."""
        for p, cap in self._wgrad_capped:
            p.opts.wgrad_target_blocks = cap if on else 0

    @staticmethod
    def _wgrad_bytes(p):
        """algorithmic HBM bytes of a weight-gradient launch: x and dy read once, dW (f32) written once"""
        b = 0
        for i in range(p.num_segments):
            s = p.seg[i]
            b += 2 * s.N * s.H * s.W * s.Cin + 2 * s.N * s.Ho * s.Wo * s.Cout
        return b + 4 * p.R * p.S * p.seg[0].Cin * p.seg[0].Cout

    def _group_wgrad_steps(self):
        """Weight-gradient launches of layers with IDENTICAL geometry become one rn_conv2d_nhwc_wgrad_group call (the
        eight head-tower layers, the 3x3 layers of a ResNet stage: up to 8 per call), issued where the LAST of them
        stood in the backward order — nothing but the optimizer reads a weight gradient, and dy / the saved activation
        of the earlier layers are static buffers nobody writes again.  One launch over (layer, co tile, ci tile) tiles
        needs 1/n of the split-K pixel chunks per layer: every workgroup writes its whole 288 KB accumulator as a partial
        tile, ~75 MB per launch however small the layer.  RNET_WGRAD_GROUP=0 keeps one launch per layer."""
        lib = self.lib
        self.wgrad_groups = []
        if os.environ.get("RNET_WGRAD_GROUP", "1") == "0":
            return
        by_sig = {}
        for i, fn in enumerate(self.bwd_steps):
            item = getattr(fn, "wgrad_item", None)
            if item is None:
                continue
            p = item[0]
            sig = (p.R, p.S, p.stride_h, p.stride_w, p.pad_top, p.pad_left, p.num_segments, bytes(p.opts),
                   tuple((s.N, s.H, s.W, s.Cin, s.Ho, s.Wo, s.Cout, s.dy_pix_stride, s.x_pix_stride)
                         for s in (p.seg[k] for k in range(p.num_segments))))
            by_sig.setdefault(sig, []).append(i)
        drop, replace = set(), {}
        for sig, idxs in by_sig.items():
            for lo in range(0, len(idxs), 8):
                grp = idxs[lo:lo + 8]
                if len(grp) < 2:
                    continue
                items = [self.bwd_steps[i].wgrad_item for i in grp]
                arr = (ctypes.POINTER(_C.WgradProblem) * len(grp))(*[ctypes.pointer(it[0]) for it in items])
                if lib.rn_wgrad_group_fused(arr, len(grp)) != 1:
                    continue
                # Groups the halo kernel does not serve — the 1x1 layers of a ResNet stage, five / six of one geometry, which
                # rn_conv2d_nhwc_wgrad_group runs as segments of one partial-tile launch since round 6 — are NOT grouped by
                # default: same-box A/B (tools/probes/ab_env_r06.sh, profiles/r06_ab/wgrad_group_1x1.txt) 28.11 / 28.32 / 28.47
                # grouped against 28.10 / 28.14 / 28.22 ms per step.  A group is issued where its LAST layer stood, which
                # moves ~40 small launches' work to the end of the weight-gradient stream — the stream that already ends
                # 0.5 ms after the main one.  RNET_WGRAD_GROUP=all groups them (the library path stays tested).
                if os.environ.get("RNET_WGRAD_GROUP", "halo") != "all" and \
                        lib.rn_wgrad_kernel_id(ctypes.byref(items[0][0])) != 2:
                    continue
                nws = lib.rn_wgrad_group_workspace_bytes(arr, len(grp))
                caps = [int(it[0].opts.wgrad_target_blocks) for it in items]
                if any(caps):          # the uncapped plan (set_wgrad_cap(False)) must fit too
                    for it in items:
                        it[0].opts.wgrad_target_blocks = 0
                    nws = max(nws, lib.rn_wgrad_group_workspace_bytes(arr, len(grp)))
                    for it, c in zip(items, caps):
                        it[0].opts.wgrad_target_blocks = c
                if os.environ.get("RNET_AB_WORKSPACES") == "1":   # tools/ab_step.py switches kernel families between rounds
                    nws = max([nws] + [int(it[2].numel()) for it in items])
                ws = torch.empty((max(nws, 256),), dtype=torch.uint8, device=self.dev)
                dws = _C.ptr_array([it[1] for it in items])
                flw = sum(it[3] for it in items)
                writes = [w for i in grp for w in self.bwd_steps[i].writes]
                a = (arr, len(grp), dws, 0.0, ws.data_ptr(), ws.numel())
                kname = ("wgrad_kernel (128x128 per-tap tiles)", "wgrad_big_kernel (256x256 per-tap tiles)",
                         "wgrad_halo_kernel")[max(lib.rn_wgrad_kernel_id(ctypes.byref(items[0][0])), 0)]
                wname = f"{kname} ({len(grp)} layers per launch) + wgrad_reduce_kernel"

                byw = sum(self._wgrad_bytes(it[0]) for it in items)
                lname = "wgrad:" + "+".join(n[len("wgrad:"):] for n, q in self.wgrad_launches
                                            if any(q is it[0] for it in items))

                def wgrad_group(st, a=a, flw=flw, wname=wname, byw=byw, lname=lname):
                    prof, lprof = self.wgrad_profile, self.layer_profile
                    if prof is None and lprof is None:
                        _C.check(lib.rn_conv2d_nhwc_wgrad_group(*a, st), "rn_conv2d_nhwc_wgrad_group")
                        return
                    cur = torch.cuda.current_stream(self.dev)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    _C.check(lib.rn_conv2d_nhwc_wgrad_group(*a, st), "rn_conv2d_nhwc_wgrad_group")
                    e1.record(cur)
                    if prof is not None:
                        prof.append((e0, e1, flw, wname))
                    if lprof is not None:
                        lprof.append((e0, e1, lname, flw, byw, wname))
                self._keep += [arr, dws, ws]
                old_ws = {id(it[2]) for it in items}
                self._keep = [k for k in self._keep if id(k) not in old_ws]      # the per-layer workspaces are not needed
                replace[grp[-1]] = self._side(wgrad_group, writes=writes)
                drop.update(grp[:-1])
                self.wgrad_groups.append([it[0] for it in items])
        self.bwd_steps = [replace.get(i, fn) for i, fn in enumerate(self.bwd_steps) if i not in drop]

    def _plan_conv_backward(self, ops, mark):
        lib, B = self.lib, self.B
        if not self._conv_trainable(ops[0]) and not any(self.requires.get(o["inp"]) for o in ops):
            return
        if not self._conv_trainable(ops[0]):
            raise NotImplementedError("backward through a frozen conv that sits above trainable layers")
        live_bn = bool(self._bn_trainable(ops[0]))
        # (a) gradient wrt the conv output
        if live_bn:
            pb, sums, bsums, ws, dys, _ = self.bn_groups[ops[0]["out"]]
            for i, op in enumerate(ops):
                s = pb.seg[i]
                s.dz = self.grad[op["out"]].data_ptr()
                if op.get("residual") and self.requires.get(op["residual"]):
                    res = op["residual"]
                    s.dres = self.grad[res].data_ptr()
                    s.dres_accumulate = 0 if mark(res) else 1
            prb = ctypes.byref(pb)
            ws = self.bn_bwd_ws.get(id(pb), ws)   # stage 1 already written there by the dgrad launch that produced dz

            def run(st, prb=prb, ws=ws, bsums=bsums, pb=pb):
                self._bn_pass("bn_bwd_reduce", pb, lambda: _C.check(lib.rn_bn_bwd_reduce(prb, _C.ptr(ws), ws.numel(), st),
                                                                     "rn_bn_bwd_reduce"))
                if self.sync_bn:
                    self._allreduce_small(bsums)
                self._bn_pass("bn_bwd_apply", pb, lambda: _C.check(lib.rn_bn_bwd_apply(prb, st), "rn_bn_bwd_apply"))
            run.writes = [op["bn"] + sfx for op in ops for sfx in ("/gamma", "/beta")]
            self.bwd_steps.append(run)
            dy_of = {op["out"]: dys[i] for i, op in enumerate(ops)}
        else:
            dy_of = {}
            for op in ops:
                if op["out_dtype"] == "f32":      # prediction convs: loss gradient arrives in fp32
                    shp = list(self.t[op["out"]].shape)
                    shp[3] = (shp[3] + 63) // 64 * 64     # K dimension of the dgrad GEMM: pad 36/720 -> 64/768
                    dyb = torch.zeros(shp, dtype=self.h16, device=self.dev)
                    dy_of[op["out"]] = dyb
                else:
                    dyb = torch.empty_like(self.t[op["out"]])
                    dy_of[op["out"]] = dyb
                    a = (self.grad[op["out"]].data_ptr(), self.t[op["out"]].data_ptr(), dyb.data_ptr(),
                         dyb.numel(), _C.ACT_IDS[op["act"]])
                    self.bwd_steps.append(lambda st, a=a: _C.check(lib.rn_act_bwd(*a, st), "rn_act_bwd"))
            self._keep += list(dy_of.values())
        self.dy_of = getattr(self, "dy_of", {})
        self.dy_of.update(dy_of)
        # (b) weight / bias gradients, one problem per distinct (shared) conv layer
        by_conv = {}
        for op in ops:
            by_conv.setdefault(op["conv"], []).append(op)
        for cname, cops in by_conv.items():
            c = self.g.convs[cname]
            p = _C.WgradProblem()
            p.opts = self.launch_opts
            p.R = p.S = c["k"]
            p.stride_h = p.stride_w = c["stride"]
            p.pad_top = p.pad_left = cops[0]["pad"]
            p.num_segments = len(cops)
            for i, op in enumerate(cops):
                x, dy = self._src(op["inp"]), dy_of[op["out"]]
                s = p.seg[i]
                s.x, s.dy = x.data_ptr(), dy.data_ptr()
                s.N, s.H, s.W, s.Cin, s.Ho, s.Wo, s.Cout = B, x.shape[1], x.shape[2], c["cin"], dy.shape[1], dy.shape[2], c["cout"]
                s.dy_pix_stride = dy.shape[3]
            nws = lib.rn_wgrad_workspace_bytes(ctypes.byref(p))
            # Two-stream backward: a weight-gradient launch is capped to ~2/3 of the CUs (rn_launch_opts.wgrad_target_blocks).
            # Its persistent workgroups own a CU for hundreds of microseconds; when they cover the whole chip every
            # main-stream launch — the critical path — queues behind them.  Measured in one process (tools/ab_step.py):
            # 31.40 -> 30.48 ms and 32.45 -> 31.34 ms per step on two boxes with 176 workgroups for the 256-wide kernels
            # and 256 (two per CU) for the 128-tile kernel; 128 / 192 is already slower again.  RNET_WGRAD_CUS=0: no cap.
            if self._wgrad_cap and not p.opts.wgrad_target_blocks:
                kid = lib.rn_wgrad_kernel_id(ctypes.byref(p))
                p.opts.wgrad_target_blocks = self._wgrad_cap[0] if kid in (1, 2) else self._wgrad_cap[1]
                nws = max(nws, lib.rn_wgrad_workspace_bytes(ctypes.byref(p)))   # either plan fits: set_wgrad_cap() may lift it
                self._wgrad_capped.append((p, int(p.opts.wgrad_target_blocks)))
            if os.environ.get("RNET_AB_WORKSPACES") == "1":   # tools/ab_step.py switches kernel families between timed rounds
                for alt in (1, 3):
                    q = _C.WgradProblem.from_buffer_copy(p)
                    q.opts.wgrad_kernel = alt
                    nws = max(nws, lib.rn_wgrad_workspace_bytes(ctypes.byref(q)))
            ws = torch.empty((max(nws, 256),), dtype=torch.uint8, device=self.dev)
            dw = self._pview(c.get("kvar", cname + "/kernel"), self.G)
            self._keep += [p, ws]
            self.wgrad_launches.append(("wgrad:" + cname, p))
            a = (ctypes.byref(p), dw.data_ptr(), 0.0, ws.data_ptr(), ws.numel())
            # algorithmic FLOPs of the layer's weight gradient: 2 * pixels * k*k * Cin * Cout over the segments
            flw = sum(2 * B * p.seg[i].Ho * p.seg[i].Wo * c["k"] * c["k"] * c["cin"] * c["cout"] for i in range(len(cops)))
            wname = ("wgrad_kernel (128x128 per-tap tiles)", "wgrad_big_kernel (256x256 per-tap tiles)",
                     "wgrad_halo_kernel")[max(lib.rn_wgrad_kernel_id(ctypes.byref(p)), 0)] + " + wgrad_reduce_kernel"

            byw = self._wgrad_bytes(p)

            def wgrad(st, a=a, flw=flw, wname=wname, byw=byw, lname="wgrad:" + cname):
                prof, lprof = self.wgrad_profile, self.layer_profile
                if prof is None and lprof is None:
                    _C.check(lib.rn_conv2d_nhwc_wgrad(*a, st), "rn_conv2d_nhwc_wgrad")
                    return
                # bench.py: HIP events on the stream the launch goes to (the side stream in the two-stream backward)
                cur = torch.cuda.current_stream(self.dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                _C.check(lib.rn_conv2d_nhwc_wgrad(*a, st), "rn_conv2d_nhwc_wgrad")
                e1.record(cur)
                if prof is not None:
                    prof.append((e0, e1, flw, wname))
                if lprof is not None:
                    lprof.append((e0, e1, lname, flw, byw, wname))
            step = self._side(wgrad, writes=[c.get("kvar", cname + "/kernel")])
            step.wgrad_item = (p, dw, ws, flw)      # _group_wgrad_steps may merge it with same-shape layers
            self.bwd_steps.append(step)
            if c["bias"] and not (os.environ.get("RNET_DEBUG_SKIP_BIAS_GRAD") == "1"):   # (debug switch: timing probe only)
                # bias gradient = column sums of dy over every segment (two-stage reduction kernel)
                pb2 = _C.BnProblem()
                pb2.num_segments, pb2.act, pb2.bessel, pb2.eps, pb2.momentum, pb2.count_scale = len(cops), 0, 0, 0.0, 0.0, 1.0
                cw = dy_of[cops[0]["out"]].shape[3]    # channel width of dy (padded for prediction convs)
                bs = torch.zeros((len(cops), 2, cw), dtype=torch.float32, device=self.dev)
                for i, op in enumerate(cops):
                    dy = dy_of[op["out"]]
                    s = pb2.seg[i]
                    s.y, s.sums = dy.data_ptr(), bs[i].data_ptr()
                    s.P, s.C = dy.shape[0] * dy.shape[1] * dy.shape[2], cw
                # A conv in front of a live BatchNorm: its dy is written by rn_bn_bwd_apply, which can sum the columns of
                # what it stores on the way out (rn_bn_segment.dy_colsum_partial) — stage 1 of this reduction then never
                # reads the dy tensor again (FPN + head-tower convs: 17 launches, ~1 ms of side-stream HBM reads per step;
                # same-process A/B of the step with / without ANY bias-gradient launch: 29.73 / 29.50 ms).
                # RNET_FUSE_BIAS_GRAD=0 keeps the separate pass.
                fused_cs = False
                if live_bn and os.environ.get("RNET_FUSE_BIAS_GRAD", "1") != "0":
                    seg_of = [ops.index(op) for op in cops]
                    chunks = [lib.rn_bn_bwd_colsum_chunks(ctypes.byref(pb), j) for j in seg_of]
                    if all(ch > 0 for ch in chunks):
                        fused_cs = True
                        for i, ch in enumerate(chunks):
                            pb2.seg[i].ext_chunks = ch
                ws2 = torch.zeros((max(lib.rn_bn_workspace_bytes(ctypes.byref(pb2)), 256),), dtype=torch.uint8,
                                  device=self.dev)
                if fused_cs:
                    for i, j in enumerate(seg_of):
                        pb.seg[j].dy_colsum_partial = ws2.data_ptr() + lib.rn_bn_partial_offset_bytes(ctypes.byref(pb2), i)
                db = self._pview(cname + "/bias", self.G)
                self._keep += [pb2, bs, ws2]

                def bias_grad(st, pr=ctypes.byref(pb2), ws2=ws2, bs=bs, db=db, n=c["cout"], rows=len(cops), stride=2 * cw):
                    _C.check(lib.rn_bn_stats(pr, _C.ptr(ws2), ws2.numel(), st), "bias colsum")
                    # sum of the per-level column sums (row 0 of every [2][cw] block), levels in order
                    _C.check(lib.rn_reduce_rows_f32(_C.ptr(bs), rows, stride, n, 0.0, _C.ptr(db), st), "bias grad")
                self.bwd_steps.append(self._side(bias_grad, writes=[cname + "/bias"]))
        # (c) data gradients.  Segments of one launch must write distinct gradient buffers (both
        # heads read the same pyramid level): split the group into launches with unique inputs.
        need = [op for op in ops if self.requires.get(op["inp"])]
        launches = []
        for op in need:
            for sub in launches:
                if all(o["inp"] != op["inp"] for o in sub):
                    sub.append(op)
                    break
            else:
                launches.append([op])
        packs = {}
        for sub in launches:
            self._plan_dgrad_launch(sub, dy_of, mark, packs)

    def _plan_dw_backward(self, ops, mark):
        """depthwise conv (+BN+act) backward: BN backward -> dy; weight gradient (summed over the levels of a
        shared separable conv); data gradient = the forward kernel on (zero-upsampled) dy with the
        tap-reversed filter, accumulating into gradient buffers that already hold a contribution."""
        lib, B = self.lib, self.B
        live_bn = bool(self._bn_trainable(ops[0]))
        if live_bn:
            pb, sums, bsums, ws, dys, _ = self.bn_groups[ops[0]["out"]]
            for i, op in enumerate(ops):
                pb.seg[i].dz = self.grad[op["out"]].data_ptr()
            prb = ctypes.byref(pb)

            def run(st, prb=prb, ws=ws, bsums=bsums, pb=pb):
                self._bn_pass("bn_bwd_reduce", pb, lambda: _C.check(lib.rn_bn_bwd_reduce(prb, _C.ptr(ws), ws.numel(), st),
                                                                     "rn_bn_bwd_reduce"))
                if self.sync_bn:
                    self._allreduce_small(bsums)
                self._bn_pass("bn_bwd_apply", pb, lambda: _C.check(lib.rn_bn_bwd_apply(prb, st), "rn_bn_bwd_apply"))
            run.writes = [op["bn"] + sfx for op in ops for sfx in ("/gamma", "/beta")]
            self.bwd_steps.append(run)
            dy_of = {op["out"]: dys[i] for i, op in enumerate(ops)}
        else:
            dy_of = {op["out"]: self.grad[op["out"]] for op in ops}
        by_layer = {}
        for op in ops:
            by_layer.setdefault(op["dw"], []).append(op)
        for dname, dops in by_layer.items():
            d = self.g.dws[dname]
            p = _C.DwProblem()
            p.k, p.stride, p.pad_top, p.pad_left, p.act = d["k"], d["stride"], dops[0]["pad_top"], dops[0]["pad_left"], 0
            p.num_segments = len(dops)
            for i, op in enumerate(dops):
                x, dy = self._src(op["inp"]), dy_of[op["out"]]
                s = p.seg[i]
                s.x, s.y = x.data_ptr(), dy.data_ptr()
                s.N, s.H, s.W, s.C, s.Ho, s.Wo = B, x.shape[1], x.shape[2], d["C"], dy.shape[1], dy.shape[2]
            ws = torch.empty((max(lib.rn_depthwise_wgrad_workspace_bytes(ctypes.byref(p)), 256),), dtype=torch.uint8,
                             device=self.dev)
            self._keep += [p, ws]
            a = (ctypes.byref(p), self._pview(d["kvar"], self.G).data_ptr(), ws.data_ptr(), ws.numel())
            self.bwd_steps.append(self._side(lambda st, a=a: _C.check(lib.rn_depthwise_conv2d_nhwc_wgrad(*a, st),
                                                                     "dw wgrad"), writes=[d["kvar"]]))
        need = [op for op in ops if self.requires.get(op["inp"])]
        launches = []
        for op in need:
            for sub in launches:
                if all(o["inp"] != op["inp"] for o in sub):
                    sub.append(op)
                    break
            else:
                launches.append([op])
        flips = {}
        for sub in launches:
            d0 = self.g.dws[sub[0]["dw"]]
            k, stride = d0["k"], d0["stride"]
            p = _C.DwProblem()
            p.k, p.stride, p.act, p.num_segments = k, 1, 0, len(sub)
            p.pad_top, p.pad_left = k - 1 - sub[0]["pad_top"], k - 1 - sub[0]["pad_left"]
            ups = []
            for i, op in enumerate(sub):
                d = self.g.dws[op["dw"]]
                if op["dw"] not in flips:
                    buf = torch.empty((k * k, d["C"]), dtype=self.h16, device=self.dev)
                    flips[op["dw"]] = buf
                    off, _ = self.p_off[d["kvar"]]
                    self.dw_flip_packs.append((self.P.data_ptr() + 4 * off, k, d["C"], buf))
                dy = dy_of[op["out"]]
                x = self._src(op["inp"])
                H, W = x.shape[1], x.shape[2]
                if stride == 2:
                    up = torch.empty((B, H, W, d["C"]), dtype=self.h16, device=self.dev)
                    ups.append((dy.data_ptr(), up.data_ptr(), B, dy.shape[1], dy.shape[2], d["C"], H, W))
                    self._keep.append(up)
                    src = up
                elif stride == 1:
                    src = dy
                else:
                    raise NotImplementedError("stride > 2")
                gbuf = self._gradbuf(op["inp"])
                first = mark(op["inp"] if op["inp"] not in self.bal_src else "bal:" + op["inp"])
                s = p.seg[i]
                s.x, s.w, s.y = src.data_ptr(), flips[op["dw"]].data_ptr(), gbuf.data_ptr()
                s.scale, s.shift = None, None
                s.residual = None if first else gbuf.data_ptr()
                s.N, s.H, s.W, s.C, s.Ho, s.Wo = B, H, W, d["C"], H, W
            self._keep.append(p)

            def dgrad(st, p=p, ups=ups):
                for u in ups:
                    _C.check(lib.rn_upsample_zero2x(*u, st), "rn_upsample_zero2x")
                _C.check(lib.rn_depthwise_conv2d_nhwc_fwd(ctypes.byref(p), st), "dw dgrad")
            self.bwd_steps.append(dgrad)

    def _consumers(self, name):
        """how many ops read tensor `name` (conv / depthwise / pooling inputs, residual inputs, top-down / balance
        members)"""
        if not hasattr(self, "_consumer_count"):
            cnt = {}
            for op in self.ops:
                for key, val in op.items():
                    if key in ("op", "out", "outs", "conv", "bn", "dw", "act", "group", "out_dtype", "kvar"):
                        continue
                    for v in (val if isinstance(val, (list, tuple)) else [val]):
                        if isinstance(v, str) and v in self.tensors:
                            cnt[v] = cnt.get(v, 0) + 1
            self._consumer_count = cnt
        return self._consumer_count.get(name, 0)

    def _bn_bwd_fusable(self, name):
        """(rn_bn_problem, segment) of the BatchNorm + ReLU layer that produced `name` when stage 1 of its backward
        reduction can run in the epilogue of the ONE data-gradient launch that writes its dz: trainable BatchNorm,
        relu, no residual input, no drop_connect factors, a single-segment group, a single consumer."""
        if not self.fuse_bn_bwd or name in self.bal_src or self._consumers(name) != 1:
            return None
        if not hasattr(self, "_bn_of_tensor"):
            self._bn_of_tensor = {}
            for pb, _, _, _, _, gops in self.bn_groups.values():
                for i, o in enumerate(gops):
                    self._bn_of_tensor[o["out"]] = (pb, i, o, len(gops))
        hit = self._bn_of_tensor.get(name)
        if hit is None:
            return None
        pb, i, o, nseg = hit
        if nseg != 1 and not self.fuse_bn_bwd_groups:
            return None
        if (o["op"] != "conv" or o.get("act") != "relu" or o.get("residual") or
                o.get("survival") is not None or not self._bn_trainable(o) or pb.seg[i].sample_scale):
            return None
        return pb, i

    def _plan_dgrad_launch(self, need, dy_of, mark, packs):
        lib, B = self.lib, self.B
        c0 = self.g.convs[need[0]["conv"]]
        k, stride = c0["k"], c0["stride"]
        p = _C.attach_splitk_workspace(_C.ConvProblem(), self.splitk_ws)
        p.opts = self.launch_opts
        p.R = p.S = k
        p.stride_h = p.stride_w = 1
        p.pad_top = p.pad_left = k - 1 - need[0]["pad"]
        p.act, p.out_dtype, p.num_segments = _C.RN_ACT_NONE, _C.RN_DT_BF16, len(need)
        ups = []
        # 1x1 / stride 2 (projection shortcuts): dx is non-zero only at the even positions, so the GEMM runs on dy
        # as it is (a quarter of the pixels of the zero-upsampled form) and rn_scatter_add2x puts the rows in place
        lowres = k == 1 and stride == 2 and need[0]["pad"] == 0 and os.environ.get("RNET_DGRAD_LOWRES", "1") != "0"
        scatters = []
        # 3x3 / stride 2 / pad 1 on even inputs (the first block of ResNet stages 2-4): sub-pixel form — one 2x2
        # stride-1 conv of dy with 4*Cin phase-major output channels + a depth-to-space, 16 tap products per dy pixel
        # instead of the 36 (9 useful) of the zero-upsampled form (rn_dgrad_pack.pad_ == 1)
        subpixel = (k == 3 and stride == 2 and need[0]["pad"] == 1 and os.environ.get("RNET_DGRAD_SUBPIXEL", "1") != "0"
                    and all(self._src(o["inp"]).shape[1] % 2 == 0 and self._src(o["inp"]).shape[2] % 2 == 0
                            and self.g.convs[o["conv"]]["cin"] % 8 == 0 for o in need))   # rn_depth_to_space2x: C % 8
        if subpixel:
            p.R = p.S = 2
            p.pad_top = p.pad_left = 0
        d2s = []
        plain_first = []   # per segment of the plain form: this launch is the first writer of its gradient buffer
        for i, op in enumerate(need):
            c = self.g.convs[op["conv"]]
            dy = dy_of[op["out"]]
            cw = dy.shape[3]
            cwp = lib.rn_conv_cin_pad(cw)     # K of the dgrad GEMM, zero padded in the packed weights only
            if op["conv"] not in packs:
                if subpixel:
                    buf = torch.empty((lib.rn_conv_cout_pad(4 * c["cin"]), 2, 2, cwp), dtype=self.h16, device=self.dev)
                else:
                    buf = torch.empty((lib.rn_conv_cout_pad(c["cin"]), k, k, cwp), dtype=self.h16, device=self.dev)
                packs[op["conv"]] = buf
                off, _ = self.p_off[c.get("kvar", op["conv"] + "/kernel")]
                self.dgrad_packs.append((self.P.data_ptr() + 4 * off, k, c["cin"], c["cout"], cwp, buf, 1 if subpixel else 0))
            x = self._src(op["inp"])
            H, W = x.shape[1], x.shape[2]
            if subpixel:
                gbuf = self._gradbuf(op["inp"])
                first = mark(op["inp"] if op["inp"] not in self.bal_src else "bal:" + op["inp"])
                tmp = torch.empty((B, dy.shape[1], dy.shape[2], 4 * c["cin"]), dtype=self.h16, device=self.dev)
                self._keep.append(tmp)
                d2s.append((tmp.data_ptr(), gbuf.data_ptr(), B, dy.shape[1], dy.shape[2], c["cin"], 0 if first else 1))
                s = p.seg[i]
                s.x, s.w, s.y = dy.data_ptr(), packs[op["conv"]].data_ptr(), tmp.data_ptr()
                s.scale, s.shift, s.residual = None, None, None
                s.N, s.H, s.W, s.Cin, s.pix_stride = B, dy.shape[1], dy.shape[2], cw, cw
                s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], 4 * c["cin"]
                continue
            if lowres:
                gbuf = self._gradbuf(op["inp"])
                first = mark(op["inp"] if op["inp"] not in self.bal_src else "bal:" + op["inp"])
                tmp = torch.empty((B, dy.shape[1], dy.shape[2], c["cin"]), dtype=self.h16, device=self.dev)
                self._keep.append(tmp)
                scatters.append((tmp.data_ptr(), gbuf.data_ptr(), B, dy.shape[1], dy.shape[2], c["cin"], H, W,
                                 0 if first else 1))
                s = p.seg[i]
                s.x, s.w, s.y = dy.data_ptr(), packs[op["conv"]].data_ptr(), tmp.data_ptr()
                s.scale, s.shift, s.residual = None, None, None
                s.N, s.H, s.W, s.Cin, s.pix_stride = B, dy.shape[1], dy.shape[2], cw, cw
                s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], c["cin"]
                continue
            if stride == 2:
                up = torch.empty((B, H, W, cw), dtype=self.h16, device=self.dev)
                ups.append((dy.data_ptr(), up.data_ptr(), B, dy.shape[1], dy.shape[2], cw, H, W))
                self._keep.append(up)
                src = up
            elif stride == 1:
                src = dy
            else:
                raise NotImplementedError("stride > 2")
            gbuf = self._gradbuf(op["inp"])
            first = mark(op["inp"] if op["inp"] not in self.bal_src else "bal:" + op["inp"])
            plain_first.append(first and stride == 1)
            s = p.seg[i]
            s.x, s.w, s.y = src.data_ptr(), packs[op["conv"]].data_ptr(), gbuf.data_ptr()
            s.scale, s.shift = None, None
            s.residual = None if first else gbuf.data_ptr()
            s.N, s.H, s.W, s.Cin, s.pix_stride = B, src.shape[1], src.shape[2], cw, cw
            s.Ho, s.Wo, s.Cout = H, W, c["cin"]
        # stage 1 of the BatchNorm backward reduction of the layers whose dz this launch writes (all segments or none)
        bn_fused = 0
        hits = [self._bn_bwd_fusable(op["inp"]) for op in need] if len(plain_first) == len(need) and all(plain_first) else []
        # all segments of a BatchNorm problem or none (rn_bn_bwd_reduce): a problem is switched over once the launches
        # planned so far write dz of EVERY one of its segments, each exactly once — one launch for a bottleneck layer,
        # one launch over both heads x five levels for head-tower depths 0-2, the two prediction convs' data gradients
        # together for depth 3.  Until then the assignments wait in self._bn_bwd_pending; the rn_conv_problem structs
        # are read at launch time, so they can still be patched when the last segment turns up.
        if hits and all(h is not None for h in hits):
            for i, (op, (pb, j)) in enumerate(zip(need, hits)):
                pend = self._bn_bwd_pending.setdefault(id(pb), {"pb": pb, "seg": {}})
                if j in pend["seg"]:          # a second writer: not a single-consumer layer after all
                    pend["dead"] = True
                pend["seg"][j] = (p, i, op["inp"], None)
            for key in {id(pb) for pb, _ in hits}:
                pend = self._bn_bwd_pending[key]
                pb = pend["pb"]
                if pend.get("dead") or pend.get("done") or sorted(pend["seg"]) != list(range(pb.num_segments)):
                    continue
                pend["done"] = True
                for j, (pc_, i_, _, _) in pend["seg"].items():
                    pb.seg[j].ext_chunks_bwd = lib.rn_conv_bn_row_blocks(ctypes.byref(pc_), i_)
                wsb = torch.zeros((max(lib.rn_bn_workspace_bytes(ctypes.byref(pb)), 256),), dtype=torch.uint8, device=self.dev)
                self.bn_bwd_ws[key] = wsb
                for j, (cp, ci, name, _) in pend["seg"].items():
                    sg = cp.seg[ci]
                    sg.bn_partial = wsb.data_ptr() + lib.rn_bn_bwd_partial_offset_bytes(ctypes.byref(pb), j)
                    sg.bn_bwd_y = self.raw[name].data_ptr()
                    sg.bn_bwd_fwd = pb.seg[j].fwd
                    self.bn_bwd_fused.append(name)
                    if id(cp) in self._algo:    # an earlier launch: add the bytes of y its epilogue now reads
                        fl0, by0 = self._algo[id(cp)]
                        self._algo[id(cp)] = (fl0, by0 + 2 * int(pb.seg[j].P) * int(pb.seg[j].C))
                    else:
                        bn_fused += 2 * int(pb.seg[j].P) * int(pb.seg[j].C)
        self._keep.append(p)
        self.conv_launches.append(("dgrad:" + (need[0].get("group") or need[0]["out"]), p))
        fl = by = 0
        for op in need:     # the layer's own MACs and tensors: dy [B,Ho,Wo,Cout] in, dx [B,H,W,Cin] out (+ accumulate read)
            c = self.g.convs[op["conv"]]
            Ho, Wo = self.tensors[op["out"]][:2]
            x = self._src(op["inp"])
            fl += 2 * B * Ho * Wo * c["k"] * c["k"] * c["cin"] * c["cout"]
            by += 2 * B * Ho * Wo * c["cout"] + 2 * B * x.shape[1] * x.shape[2] * c["cin"] + 2 * c["k"] * c["k"] * c["cin"] * c["cout"]
        by += sum(2 * B * self._src(o["inp"]).shape[1] * self._src(o["inp"]).shape[2] * self.g.convs[o["conv"]]["cin"]
                  for i, o in enumerate(need) if p.seg[i].residual)
        by += bn_fused
        self._algo[id(p)] = (fl, by)

        def dgrad(st, p=p, ups=ups, scatters=scatters, d2s=d2s):
            for u in ups:
                _C.check(lib.rn_upsample_zero2x(*u, st), "rn_upsample_zero2x")
            self._launch_conv(p, st, "dgrad")
            for sc in scatters:
                _C.check(lib.rn_scatter_add2x(*sc, st), "rn_scatter_add2x")
            for ds in d2s:
                _C.check(lib.rn_depth_to_space2x(*ds, st), "rn_depth_to_space2x")
        self.bwd_steps.append(dgrad)

    # ---- one training step -----------------------------------------------------------------------------
    def refresh_dgrad_weights(self, st):
        lib = self.lib
        if self.dgrad_packs and os.environ.get("RNET_BATCH_DGRAD_PACK", "1") == "0":   # A/B switch (tools/)
            for (mptr, k, cin, cout, cw, buf, mode) in self.dgrad_packs:
                if mode:
                    raise RuntimeError("RNET_BATCH_DGRAD_PACK=0 has no sub-pixel packing: set RNET_DGRAD_SUBPIXEL=0 with it")
                _C.check(lib.rn_pack_conv_weight_dgrad(mptr, k, k, cin, cout, cw, buf.data_ptr(), st), "pack dgrad")
        elif self.dgrad_packs:
            if getattr(self, "_dgrad_pack_items", None) is None:   # descriptors are static: build them once
                arr = (_C.DgradPack * len(self.dgrad_packs))()
                for i, (mptr, k, cin, cout, cw, buf, mode) in enumerate(self.dgrad_packs):
                    arr[i].w_ohwi, arr[i].w_packed = mptr, buf.data_ptr()
                    arr[i].R, arr[i].S, arr[i].Cin, arr[i].Cout, arr[i].Cout_pad, arr[i].pad_ = k, k, cin, cout, cw, mode
                self._dgrad_pack_items = arr
            _C.check(lib.rn_pack_conv_weight_dgrad_batch(self._dgrad_pack_items, len(self.dgrad_packs), st),
                     "pack dgrad")
        for (mptr, k, C, buf) in self.dw_flip_packs:
            _C.check(lib.rn_pack_depthwise_weight_flip(mptr, k, C, buf.data_ptr(), st), "pack dw flip")

    def draw_drop_connect(self):
        """binary = floor(survival_prob + U[0,1)); factor = binary / survival_prob, per image (:107-112)."""
        u = torch.rand(self.dc_all.shape, generator=self.dc_generator, device=self.dev, dtype=torch.float32)
        torch.div(torch.floor(u.add_(self.dc_p)), self.dc_p, out=self.dc_all)

    def forward(self, images, draw=True):
        st = _C.current_stream()
        if draw and self.dc_masks:
            self.draw_drop_connect()
        own = self.t["images"]
        if (images.device == own.device and images.dtype == own.dtype and images.shape == own.shape
                and images.is_contiguous()):
            # the only reader is the NHWC4 packing kernel at the head of the launch list: read the batch where it is
            # (a 157 MB device copy per step at 640x640x32 otherwise)
            self._images_ref = images
            self._images_ptr = images.data_ptr()
        else:
            own.copy_(images, non_blocking=True)
            self._images_ptr = own.data_ptr()
        for fn in self.fwd_steps:
            fn(st)
        return self.outputs

    @staticmethod
    def _side(fn, writes=()):
        """marks a backward step whose results only the optimizer reads (weight / bias gradients); `writes` = the
        variables whose gradient the step completes (bucket readiness of the overlapped all-reduce)"""
        fn.side = True
        fn.writes = list(writes)
        return fn

    def _ensure_side_stream(self):
        if self._side_stream is None:
            # a stream that really runs beside the caller's (probed: HIP's stream -> hardware-queue map depends on how
            # many streams the process created before — _C.concurrent_stream)
            probes, agree = [], None
            if self.dp_active and self.sync_bn and self.native_comm is None:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_backend(self.pg) == "nccl":
                    # the SyncBN messages hop main -> c10d's stream -> main: that stream must not share a queue with this
                    # one either (its packets would wait behind every weight gradient queued before them)
                    tiny = torch.zeros((64,), dtype=torch.float32, device=self.dev)
                    dist.all_reduce(tiny, group=self.pg)      # (c10d picks its stream at the first collective)
                    probes.append(lambda tiny=tiny: dist.all_reduce(tiny, group=self.pg))

                    def agree(ok, dist=dist):          # every rank keeps or drops the candidate together
                        v = torch.tensor([1.0 if ok else 0.0], device=self.dev)
                        dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.pg)
                        return bool(v.item() == 1.0)
            self._side_stream, self.side_stream_probed = _C.concurrent_stream(
                self.lib, self.dev, [torch.cuda.current_stream(self.dev)], probes=probes, agree=agree)
            self._side_events = [torch.cuda.Event() for _ in self.bwd_steps]
        return self._side_stream

    def _prepack_dgrad_weights(self):
        """The data-gradient weight layouts (taps flipped, [Cin][R][S][Cout]) depend only on the weights the last
        optimizer step left: repack them on the second stream while the forward pass runs instead of at the head
        of the backward pass (0.2 ms of HBM-bound work off the critical path)."""
        if not self.side_stream_on:
            return
        side = self._ensure_side_stream()
        side.wait_stream(torch.cuda.current_stream(self.dev))   # after the optimizer and the previous backward pass
        with torch.cuda.stream(side):
            self.refresh_dgrad_weights(ctypes.c_void_p(side.cuda_stream))
        self._dgrad_prepacked = True

    def loss_grad_buffers(self):
        """The dy tensors of the prediction convs (bf16 [B,H,W,padded channels]) keyed like the predictions: handed to
        RetinaNetLoss(grads_bf16=...) so that the loss kernels write the upstream gradients where backward() reads
        them (pad channels stay zero from allocation)."""
        if getattr(self, "_loss_dy", None) is None:
            self._loss_dy = {k: {lv: self.dy_of[name] for lv, name in self.g.outputs[k].items()}
                             for k in ("class-predictions", "box-predictions")}
        return self._loss_dy

    def backward(self, loss_grads):
        """loss_grads: RetinaNetLoss.grads (f32, per level), or None when the loss wrote loss_grad_buffers()
        -> parameter gradients in self.G."""
        lib, st = self.lib, _C.current_stream()
        if loss_grads is not None:   # None: the loss kernels already wrote bf16 into loss_grad_buffers()
            for key, okey in (("class-predictions", "class-predictions"), ("box-predictions", "box-predictions")):
                for lv, name in self.g.outputs[okey].items():
                    gsrc = loss_grads[key][lv]
                    dst = self.dy_of[name]
                    C = gsrc.shape[-1]
                    _C.check(lib.rn_cast_pad_f32_to_bf16(gsrc.data_ptr(), dst.data_ptr(), gsrc.numel() // C, C,
                                                         dst.shape[3], st), "cast")
        if getattr(self, "_dgrad_prepacked", False):     # train_step repacked them beside the forward pass
            torch.cuda.current_stream(self.dev).wait_stream(self._side_stream)
            self._dgrad_prepacked = False
        else:
            self.refresh_dgrad_weights(st)
        overlap = self._overlap_begin()
        if not self.side_stream_on:
            for i, fn in enumerate(self.bwd_steps):
                fn(st)
                if overlap:
                    self._overlap_after_step(i, torch.cuda.current_stream(self.dev), None)
            return
        # two streams: a side step waits (event) for everything the main stream has enqueued so far — its inputs
        # dy / x are complete at that point and are not written again before the join below — and the main stream
        # carries on with the data gradients; the optimizer (clip: global norm over every gradient) follows the join
        main = torch.cuda.current_stream(self.dev)
        side = self._ensure_side_stream()
        sst = ctypes.c_void_p(side.cuda_stream)
        fresh = False            # the side stream already waits for the newest main-stream work
        for i, fn in enumerate(self.bwd_steps):
            if getattr(fn, "side", False):
                if not fresh:
                    ev = self._side_events[i]
                    ev.record(main)
                    side.wait_event(ev)
                    fresh = True
                with torch.cuda.stream(side):
                    fn(sst)
            else:
                fn(st)
                fresh = False
            if overlap:
                self._overlap_after_step(i, main, side)
        main.wait_stream(side)

    # ---- gradient all-reduce overlapped with the backward pass -------------------------------------
    # executor.py:432-437 clips the LOCAL gradients and then sums them over the replicas; the clip factors need every
    # gradient, so a literal translation can only start the all-reduce after the whole backward pass.  Here the
    # buckets go out as the backward pass completes them, UNclipped ("optimistic": local gradients are pre-divided by
    # the replica count, so the per-tensor / global norms sit far below clipnorm after the first steps), each rank
    # keeps a copy of what it sent, and at the end one flag that rode in the last bucket says whether any rank's
    # factor was != 1.  Only then is the correction sum_r (factor_r - 1) * g_r all-reduced and added — the result is
    # sum_r factor_r * g_r, the reference's clip-then-sum.
    def _plan_buckets(self):
        bucket_bytes = int(os.environ.get("RNET_C1_BUCKET_MB", "25")) << 20
        written = {}
        for i, fn in enumerate(self.bwd_steps):
            for k in getattr(fn, "writes", ()):
                written[k] = i
        missing = [k for k in self.train_names if k not in written]
        if missing:
            raise RuntimeError(f"no backward step completes the gradient of {missing[:4]}")
        buckets, cur = [], None
        for k in self.train_names:     # arena order = forward order: the backward pass completes the tail first
            off, n = self.p_off[k]
            b0, nb = self._seg_blocks[k]
            if cur is None or (cur["end"] - cur["begin"]) * 4 >= bucket_bytes:
                cur = dict(begin=off, end=off, block_begin=b0, block_count=0, ready=-1)
                buckets.append(cur)
            cur["end"] = (off + n + 3) // 4 * 4
            cur["block_count"] += nb
            cur["ready"] = max(cur["ready"], written[k])
        buckets[0]["begin"] = 0        # the flag slots ride in the bucket that completes last
        order = sorted(range(len(buckets)), key=lambda j: (buckets[j]["ready"], -j))
        if order[-1] != 0:             # keep the invariant simple: bucket 0 goes last
            buckets[0]["ready"] = max(b["ready"] for b in buckets)
        self._buckets = buckets
        self._bucket_at = {}
        for j, bkt in enumerate(buckets):
            self._bucket_at.setdefault(bkt["ready"], []).append(j)
        for lst in self._bucket_at.values():
            lst.sort(reverse=True)     # bucket 0 after the others that become ready with the same step

    def _overlap_begin(self):
        """True when this backward pass launches the gradient all-reduce bucket by bucket (world > 1, or forced
        with RNET_C1_OVERLAP=1 for the single-replica equivalence test)."""
        mode = os.environ.get("RNET_C1_OVERLAP", "auto")
        on = self._train_step_active and (mode == "1" or (mode != "0" and self.dp_active))
        self._overlap_works = []
        self._overlap_on = on
        if not on:
            return False
        if getattr(self, "_buckets", None) is None:
            self._plan_buckets()
            # Which stream prepares a bucket and hands it to RCCL.  Round 3 used a third stream of its own; round 5 measured
            # (tools/probes/dp_overlap_trace.py, 1-rank nccl group on one MI355X, rocprofv3 kernel trace) that HIP mapped
            # it onto the SAME hardware queue as the weight-gradient stream: a bucket's "wait for the main stream" packet
            # then sat in front of weight-gradient kernels that had nothing to wait for — 2.3 ms of chip idle per step
            # against 0.7 ms, step 34.4 ms against 31.4 ms for the plain order (and 44 ms with GPU_MAX_HW_QUEUES=8) — the
            # overlap machinery cost more than the all-reduce it hides.  Now the bucket work rides on the weight-gradient
            # stream itself (most of a bucket's producers are there; it waits for the main stream's BatchNorm gamma / beta
            # gradients through one event per bucket) or, in the one-stream backward, on the main stream: no extra queue.
            # RNET_C1_STREAM=own restores the third stream (A/B on a multi-GPU node).
            self._comm_stream = None
            if os.environ.get("RNET_C1_STREAM", "side") == "own":
                self._comm_stream, _ = _C.concurrent_stream(
                    self.lib, self.dev, [torch.cuda.current_stream(self.dev)] + ([self._side_stream] if self._side_stream else []))
            self._comm_events = [(torch.cuda.Event(), torch.cuda.Event()) for _ in self._buckets]
            self.L = torch.zeros_like(self.G)      # what this rank contributed (for the clip correction)
            if self.dp_active:
                import torch.distributed as dist
                self._probe_bucket_group = True
                # its own communicator: the latency-bound SyncBN all-reduces of the main stream must not queue
                # behind a 25 MB bucket on the same RCCL stream
                self.pg_c1 = dist.new_group(backend=dist.get_backend(self.pg))
        if getattr(self, "_probe_bucket_group", False):
            self._probe_bucket_group = False
            if self.native_comm_buckets is None and not self._bucket_group_is_safe():
                # c10d's stream for the bucket group shares a hardware queue with the main stream: every bucket's "wait for
                # the weight-gradient stream" packet would stall the main stream's kernels behind it.  All ranks agreed
                # (MIN): this job keeps the plain order — all-reduce after the backward pass.
                import logging
                logging.warning("gradient-bucket overlap disabled: c10d's stream for the bucket group blocks the main stream "
                                "on this process's hardware-queue map (RNET_STREAM_PROBE=0 skips the probe)")
                self._overlap_unsafe = True
        if getattr(self, "_overlap_unsafe", False):
            self._overlap_on = False
            return False
        self._overlap_done = 0
        return True

    def _bucket_group_is_safe(self):
        """Does an async all-reduce of the bucket group, issued from the weight-gradient stream while that stream still
        waits for something, leave the main stream alone — its kernels AND the SyncBN all-reduces it issues through the
        other group (two c10d streams on one hardware queue: every SyncBN message would wait for the bucket's producers)?
        (_C.wait_blocks; collective: the ranks agree.)  c10d picks a group's stream when the group is first used, so a group
        that fails is replaced by a fresh one, three times at most."""
        import torch.distributed as dist
        side = self._side_stream
        if side is None or os.environ.get("RNET_STREAM_PROBE", "1") == "0" or dist.get_backend(self.pg_c1) != "nccl":
            return True
        main = torch.cuda.current_stream(self.dev)
        tiny = torch.zeros((64,), dtype=torch.float32, device=self.dev)
        tiny2 = torch.zeros((64,), dtype=torch.float32, device=self.dev)
        helper = torch.cuda.Stream(self.dev)
        self._keep.append(helper)
        for attempt in range(4):
            with torch.cuda.stream(side):
                dist.all_reduce(tiny, group=self.pg_c1)          # c10d picks the group's stream at its first collective
            torch.cuda.synchronize(self.dev)
            blocked = False
            probes = [lambda: _C.check(self.lib.rn_probe_spin(1, ctypes.c_void_p(main.cuda_stream)), "rn_probe_spin")]
            if self.sync_bn and self.native_comm is None:
                probes.append(lambda: dist.all_reduce(tiny2, group=self.pg))
            for probe in probes:                                 # (every probe on every rank: they may be collectives)
                works = []

                def pre(works=works):   # the group's stream now waits for the weight-gradient stream, which waits for the helper
                    with torch.cuda.stream(side):
                        works.append(dist.all_reduce(tiny, group=self.pg_c1, async_op=True))
                blocked = _C.wait_blocks(self.lib, side, probe, main, helper, pre=pre) or blocked
                with torch.cuda.stream(side):
                    for w in works:
                        w.wait()
                torch.cuda.synchronize(self.dev)
            v = torch.tensor([0.0 if blocked else 1.0], device=self.dev)
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=self.pg)
            if bool(v.item() == 1.0):
                return True
            if attempt < 3:
                self.pg_c1 = dist.new_group(backend=dist.get_backend(self.pg))
        return False

    def _overlap_after_step(self, i, main, side):
        for j in self._bucket_at.get(i, ()):
            self._launch_bucket(j, main, side)

    def _launch_bucket(self, j, main, side):
        lib, bkt, comm = self.lib, self._buckets[j], self._comm_stream
        e_main, e_side = self._comm_events[j]
        if comm is None:             # the weight-gradient stream (or the only stream) carries the bucket work
            comm = side if side is not None else main
            if side is not None:
                e_main.record(main)
                side.wait_event(e_main)
        else:
            e_main.record(main)
            comm.wait_event(e_main)
            if side is not None:
                e_side.record(side)
                comm.wait_event(e_side)
        self._bucket_stream = comm
        cst = ctypes.c_void_p(comm.cuda_stream)
        a = self._step_args
        with torch.cuda.stream(comm):
            _C.check(lib.rn_optim_clip_prepare(self.G.data_ptr(), self.P.data_ptr(), self.segs_dev.data_ptr(),
                                               self.block_seg_dev.data_ptr(), self.n_blocks, bkt["block_begin"],
                                               bkt["block_count"], a["wdc"], a["unscale"], self.L.data_ptr(),
                                               self.opt_ws.data_ptr(), self.opt_ws.numel(), cst), "rn_optim_clip_prepare")
            self._overlap_done += 1
            if self._overlap_done == len(self._buckets):
                assert j == 0
                # every gradient of this rank is final: clip factors, metrics, and the two flags into G[0:2]
                _C.check(lib.rn_optim_clip_factors(self.segs_dev.data_ptr(), self.n_segs, self.n_blocks, a["clip"],
                                                   a["alpha"], self.metrics.data_ptr(), self.G.data_ptr(),
                                                   self.opt_ws.data_ptr(), self.opt_ws.numel(), cst), "rn_optim_clip_factors")
            if self.dp_active and getattr(self, "native_comm_buckets", None) is not None:
                # rn_allreduce_bucket (rn_comm.hip): one ncclAllReduce on the bucket's stream, behind its prepare kernel
                self.native_comm_buckets.all_reduce_bucket(self.G[bkt["begin"]:bkt["end"]])
            elif self.dp_active:
                import time as _time
                import torch.distributed as dist
                t0 = _time.perf_counter()
                self._overlap_works.append(dist.all_reduce(self.G[bkt["begin"]:bkt["end"]], group=self.pg_c1,
                                                           async_op=True))
                self.bucket_host_ms = max(getattr(self, "bucket_host_ms", 0.0), (_time.perf_counter() - t0) * 1e3)
        self._overlap_last_event = torch.cuda.Event()
        self._overlap_last_event.record(comm)

    def _overlap_finish(self, optimistic_sgd=None):
        """After the join: wait for the buckets; when some rank's clip fired, all-reduce the correction.
        optimistic_sgd: callable that enqueues the SGD kernel with the device-side predicate "G[0] == 0" (no clip fired on
        any rank, no gradient non-finite: the common case).  The flag goes to pinned host memory with an asynchronous copy
        enqueued BEFORE that kernel, the host waits for the copy's event only — the kernel runs while the host decides and
        carries on enqueueing (a blocking `.item()` left the device idle for the host's wake-up and the next launches:
        0.24 - 0.38 ms per step, bench.py's extra.dp_overhead).  Returns True when the optimistic kernel applied the step;
        False when the flag fired (the kernel was a no-op: the caller runs the SGD kernel after the correction below)."""
        cur = torch.cuda.current_stream(self.dev)
        for w in self._overlap_works:
            w.wait()
        if getattr(self, "_bucket_stream", None) is not None and self._bucket_stream != cur:
            cur.wait_stream(self._bucket_stream)
        applied = False
        if getattr(self, "price_without_flag_read", False):
            fired = False                            # bench.py's extra.dp_overhead ONLY: what the host read below costs
        elif optimistic_sgd is not None:
            if getattr(self, "_flag_host", None) is None:
                self._flag_host = torch.zeros((1,), dtype=torch.float32, pin_memory=True)
                self._flag_event = torch.cuda.Event()
            self._flag_host.copy_(self.G[0:1], non_blocking=True)
            self._flag_event.record(cur)
            optimistic_sgd()                         # predicate on the device: a no-op when G[0] != 0
            self._flag_event.synchronize()
            fired = float(self._flag_host[0]) != 0.0
            applied = not fired
        else:
            fired = float(self.G[0].item()) != 0.0      # one host sync per step: the collective below is conditional
        self.clip_fired = fired
        if not fired:
            return applied
        st = _C.current_stream()
        _C.check(self.lib.rn_optim_clip_apply(self.L.data_ptr(), self.L.data_ptr(), self.segs_dev.data_ptr(),
                                              self.block_seg_dev.data_ptr(), self.n_blocks, self.opt_ws.data_ptr(),
                                              self.opt_ws.numel(), st), "rn_optim_clip_apply")
        self.L[:4].zero_()
        if self.dp_active:
            from retinanet.distribute import all_reduce_sum_bucketed
            all_reduce_sum_bucketed(self.L, 2 if self.world == 1 else self.world, self.pg)   # (forced: issue it anyway)
        self.G[4:].add_(self.L[4:])
        return False

    def optimizer_step(self, lr, momentum, clipnorm, wd_alpha, ema_decay, nesterov=False, overlapped=False):
        """weight decay + per-tensor / global clipping (executor.py:401-407) + all-reduce SUM (executor.py:436-437)
        + SGD momentum / moving average (optimizers/builder.py:45-54).  overlapped=True: backward() already sent the
        buckets (see above); only the flag check / correction and the SGD kernel are left."""
        lib, st = self.lib, _C.current_stream()
        unscale = 1.0 / self.loss_scale["scale"] if self.loss_scale else 1.0
        def sgd(skip_ptr):
            _C.check(lib.rn_optim_sgd_step(self.P.data_ptr(), self.G.data_ptr(), self.V.data_ptr(),
                                           self.E.data_ptr() if ema_decay is not None else None, self.Pbf.data_ptr(),
                                           self.segs_dev.data_ptr(), self.block_seg_dev.data_ptr(), self.n_blocks,
                                           lr, momentum, ema_decay if ema_decay is not None else 0.0, 1 if nesterov else 0,
                                           skip_ptr, st), "rn_optim_sgd_step")
        if overlapped:
            # G[0] = sum over the ranks of "a clip factor != 1 or the gradient norm is not finite" (rn_optim_clip_factors:
            # a non-finite norm makes its factor != 1, so G[0] == 0 also says every gradient is finite)
            optimistic = os.environ.get("RNET_C1_OPTIMISTIC_SGD", "1") != "0"
            applied = self._overlap_finish((lambda: sgd(self.G.data_ptr())) if optimistic else None)
            if applied:
                self.refresh_stem_pack()
                if self.loss_scale:
                    self._update_loss_scale()
                return
            skip = self.G.data_ptr() + 4 if self.loss_scale else None     # G[1]: not finite on some rank
        else:
            _C.check(lib.rn_optim_clip(self.G.data_ptr(), self.P.data_ptr(), self.segs_dev.data_ptr(), self.n_segs,
                                       self.block_seg_dev.data_ptr(), self.n_blocks, wd_alpha / self.world, wd_alpha,
                                       unscale, clipnorm if clipnorm else 0.0, self.metrics.data_ptr(),
                                       self.opt_ws.data_ptr(), self.opt_ws.numel(), st), "rn_optim_clip")
            skip = None
            if self.loss_scale:
                self.G[1:2].copy_(self.metrics[5:6])
                skip = self.G.data_ptr() + 4
            if self.dp_active:
                from retinanet.distribute import all_reduce_sum_bucketed
                all_reduce_sum_bucketed(self.G, 2 if self.world == 1 else self.world, self.pg)   # executor.py:436-437: SUM after clipping
        sgd(skip)
        self.refresh_stem_pack()
        if self.loss_scale:
            self._update_loss_scale()

    def _update_loss_scale(self):
        """tf.keras.mixed_precision.LossScaleOptimizer(dynamic=True) (optimizers/builder.py:56-64): halve the scale
        when a step's gradients were not finite (the step was dropped), double it after `growth_steps` good steps.

        The SGD kernel drops the update itself (skip pointer); what the HOST needs — the next loss scale, whether
        optimizer.iterations advances — it needs only when the NEXT step reaches its loss: the flag goes to pinned
        memory with an asynchronous copy here and is applied by _resolve_loss_scale(), which the next train_step calls
        after it has enqueued its forward pass (or finish_step(), for whoever reads the counters in between).  Reading
        it here (`.item()`) drained the queue at the end of every mixed_float16 step: the device then idled ~1 ms at the
        head of the next step until the host had launched its first kernels (tools/trace_gaps.py on configs[4])."""
        if getattr(self, "_ls_host", None) is None:
            self._ls_host = torch.zeros((1,), dtype=torch.float32, pin_memory=True)
            self._ls_event = torch.cuda.Event()
        self._ls_host.copy_(self.G[1:2], non_blocking=True)
        self._ls_event.record(torch.cuda.current_stream(self.dev))
        self._ls_pending = True

    def _resolve_loss_scale(self):
        if not getattr(self, "_ls_pending", False):
            return
        self._ls_pending = False
        self._ls_event.synchronize()
        ls = self.loss_scale
        bad = float(self._ls_host[0]) != 0.0
        ls["skipped"] = bad
        if bad:
            ls["scale"], ls["good"] = max(ls["scale"] / 2.0, 1.0), 0
            self.step_count -= 1              # a dropped step does not advance optimizer.iterations
            self.model.optimizer.iterations = self.step_count
        else:
            ls["good"] += 1
            if ls["good"] >= ls["growth_steps"]:
                ls["scale"], ls["good"] = ls["scale"] * 2.0, 0

    def finish_step(self):
        """Host-side state of the last train_step (loss scale, optimizer.iterations) brought up to date: call before
        reading them between steps (the executor does after each execution; state_dict does)."""
        self._resolve_loss_scale()

    def train_step(self, images, targets):
        """(images f32[B,H,W,3], targets from LabelEncoder.encode_batch) -> the loss dict of Executor._train_step
        (executor.py:409-441; device scalars)."""
        cfg = self.params_cfg.training
        opt = self.model.optimizer
        if self.model.loss._num_replicas() != self.world:
            raise RuntimeError(f"RetinaNetLoss sees {self.model.loss._num_replicas()} replicas, the engine {self.world}")
        if opt.dynamic_loss_scale and self.loss_scale is None:
            self.loss_scale = dict(scale=float(opt.initial_loss_scale), good=0, growth_steps=int(opt.loss_scale_growth_steps),
                                   skipped=False)
        with torch.cuda.device(self.dev):
            alpha = cfg.weight_decay_alpha if cfg.use_weight_decay else 0.0
            self._prepack_dgrad_weights()
            self._small_msgs = 0
            self._c2_local, self._c2_sent, self.c2_normalizer = None, False, None
            if self.sync_bn:    # sum(num-positives) + 1 of this rank (retinanet_loss.py:38): folded into SyncBN traffic
                # (a device kernel reads it: a host tensor, or one on another GPU, must be moved first — ADVICE r4)
                npos = targets["num-positives"].to(self.dev, torch.float32).contiguous()
                self._c2_local = torch.empty((1,), dtype=torch.float32, device=self.dev)
                _C.check(self.lib.rn_reduce_rows_f32(_C.ptr(npos), npos.numel(), 1, 1, 1.0, _C.ptr(self._c2_local),
                                                     _C.current_stream()), "num-positives + 1")
            preds = self.forward(images)
            self._c2_local = None
            self._resolve_loss_scale()      # the previous step's "gradients not finite" flag: long since on the host
            step = self.step_count
            scale = self.loss_scale["scale"] if self.loss_scale else 1.0
            self._step_args = dict(wdc=alpha / self.world, alpha=alpha, unscale=1.0 / scale,
                                   clip=float(opt.clipnorm) if opt.clipnorm else 0.0)
            # per_replica_loss = total / replicas, times the loss scale under mixed_float16 (executor.py:421-425)
            loss = self.model.loss(targets, preds, compute_grads=True, grad_scale=scale / self.world,
                                   grads_bf16=self.loss_grad_buffers(), normalizer=self.c2_normalizer)
            self._train_step_active = True
            try:
                self.backward(None)
            finally:
                self._train_step_active = False
            overlapped = self._overlap_on      # backward() sent the gradient buckets as it completed them
            if overlapped and self._overlap_done != len(self._buckets):
                raise RuntimeError("overlapped all-reduce: not every gradient bucket was launched")
            self.optimizer_step(opt.lr(step), opt.momentum, opt.clipnorm, alpha,
                                opt.ema_decay(step) if opt.use_moving_average else None, nesterov=opt.nesterov,
                                overlapped=overlapped)
            self.syncbn_messages_per_step = self._small_msgs if self.sync_bn else 0
            self.step_count += 1              # (taken back by _resolve_loss_scale when the step turns out dropped)
            opt.iterations = self.step_count
        out = dict(loss)
        out["total-loss"] = loss["weighted-loss"]                # executor.py:414-419
        if cfg.use_weight_decay:
            out["l2-regularization"] = self.metrics[3]
            out["total-loss"] = loss["weighted-loss"] + self.metrics[3]
        out["gradient-norm"] = self.metrics[0] * self.world      # executor.py:440
        out["num-anchors-matched"] = loss["num-anchors-matched"] / self.B   # executor.py:439
        return out
