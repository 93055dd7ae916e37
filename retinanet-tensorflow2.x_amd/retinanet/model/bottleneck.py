"""Cross-layer fusion of the ResNet stage-1 bottleneck blocks in inference form: one `rn_bottleneck64_fwd` launch
(csrc/rn_bneck.hip) instead of three (four) conv launches per block.

The reference block is retinanet/model/backbone/resnet.py:194-248 (`bottleneck_block`, filters = 64, strides = 1 in
block_group1): conv1x1 -> BN -> relu -> conv3x3 -> BN -> relu -> conv1x1 -> BN, + shortcut (identity, or conv1x1 + BN for the
group's first block, :220-228), relu.  With every BatchNorm folded (serving; the `resnet_initial` layers frozen by the 3x
configs, model/builder.py:28-29) the 64-channel intermediates are pure HBM traffic: the fused launch keeps them on chip.

`find_blocks(graph)` recognises the pattern structurally (kernel sizes, strides, channel counts, activations, who reads the
intermediate tensors) — it does not depend on layer names; the engines add their own eligibility test (frozen / no gradient
needed) and `Bottleneck64.supported()` asks the library about the shape.  RNET_FUSE_BOTTLENECK=0 keeps the per-layer launches."""
from __future__ import annotations

import ctypes
import os

import torch

from retinanet import _C


def _reads(op):
    """tensor names an op reads"""
    out = []
    for k in ("inp", "residual", "tensor"):
        if isinstance(op.get(k), str):
            out.append(op[k])
    for k in ("ins", "tensors"):
        if isinstance(op.get(k), (list, tuple)):
            out += [t for t in op[k] if isinstance(t, str)]
    return out


def find_blocks(g):
    """-> list of dict(name, ops (graph order), a, b, out, sc | None, x, Cx) for every 64-wide stride-1 bottleneck block"""
    if os.environ.get("RNET_FUSE_BOTTLENECK", "1") == "0":
        return []
    by_out = {o["out"]: o for o in g.ops if o["op"] == "conv"}
    readers = {}
    for o in g.ops:
        for t in _reads(o):
            readers.setdefault(t, []).append(o)
    net_outs = {n for d in getattr(g, "outputs", {}).values() for n in d.values()}
    idx = {id(o): i for i, o in enumerate(g.ops)}

    def layer(op, k, cin, cout, act, residual):
        if op is None or op.get("group") is not None or op.get("out_dtype", "bf16") != "bf16" or not op.get("bn"):
            return False
        c = g.convs[op["conv"]]
        return ((c["k"], c["stride"], c["cin"], c["cout"], bool(c["bias"])) == (k, 1, cin, cout, False)
                and op["pad"] == (k - 1) // 2 and op.get("act") == act and bool(op.get("residual")) == residual)

    blocks = []
    for out in g.ops:
        if out["op"] != "conv" or not layer(out, 1, 64, 256, "relu", True):
            continue
        b = by_out.get(out["inp"])
        if not layer(b, 3, 64, 64, "relu", False):
            continue
        a = by_out.get(b["inp"])
        if a is None:
            continue
        x = a["inp"]
        Cx = g.tensors[x][2]
        if Cx not in (64, 256) or not layer(a, 1, Cx, 64, "relu", False):
            continue
        sc = None
        if out["residual"] != x:
            sc = by_out.get(out["residual"])
            if Cx != 64 or not layer(sc, 1, 64, 256, None, False) or sc["inp"] != x:
                continue
        elif Cx != 256:
            continue
        # the intermediates must be private to the block
        inner = [a["out"], b["out"]] + ([sc["out"]] if sc else [])
        mine = {id(a), id(b), id(out)} | ({id(sc)} if sc else set())
        if any(t in net_outs or any(id(r) not in mine for r in readers.get(t, [])) for t in inner):
            continue
        ops = sorted([o for o in (sc, a, b, out) if o is not None], key=lambda o: idx[id(o)])
        blocks.append(dict(name=out["out"], ops=ops, a=a, b=b, out=out, sc=sc, x=x, Cx=Cx))
    return blocks


class Bottleneck64:
    """One fused block of one engine: packed weights, folded BatchNorm vectors (stable addresses: a captured HIP graph or a
    launch list keeps them), the launch descriptor."""

    def __init__(self, lib, g, blk, B, dev, h16, launch_opts, x_tensor, y_tensor):
        self.lib, self.g, self.blk, self.B, self.dev = lib, g, blk, int(B), dev
        H, W, Cx, _ = g.tensors[blk["x"]]
        self.H, self.W, self.Cx = H, W, Cx
        self.name = "bneck:" + blk["name"]
        self.ok = (x_tensor.dtype == h16 and x_tensor.is_contiguous() and y_tensor.is_contiguous()
                   and lib.rn_bottleneck64_supported(self.B, H, W, Cx) == 1)
        if not self.ok:
            return
        self.packed = torch.empty((lib.rn_bottleneck64_packed_bytes(Cx),), dtype=torch.uint8, device=dev)
        self.affine = torch.zeros((4 * 64 + (4 if Cx == 64 else 2) * 256,), dtype=torch.float32, device=dev)
        p = _C.Bottleneck64Problem()
        p.x, p.y, p.w_packed, p.affine = x_tensor.data_ptr(), y_tensor.data_ptr(), self.packed.data_ptr(), self.affine.data_ptr()
        p.N, p.H, p.W, p.Cx = self.B, H, W, Cx
        if launch_opts is not None:
            p.opts = launch_opts
        self.problem = p
        self._ref = ctypes.byref(p)
        layers = [blk["a"], blk["b"], blk["out"]] + ([blk["sc"]] if blk["sc"] else [])
        macs = sum(g.convs[o["conv"]]["k"] ** 2 * g.convs[o["conv"]]["cin"] * g.convs[o["conv"]]["cout"] for o in layers)
        self.flops = 2 * self.B * H * W * macs                                   # algorithmic: the layers' own MACs
        self.bytes = 2 * self.B * H * W * (Cx + 256) + 2 * macs                  # block input + output + weights

    def load(self, variables, eps):
        """(re)pack the block's kernels and refold its BatchNorms from a name -> f32 tensor dict (Keras layouts)"""
        lib, g, blk = self.lib, self.g, self.blk
        kern = lambda op: variables[g.convs[op["conv"]].get("kvar", op["conv"] + "/kernel")].to(self.dev, torch.float32).contiguous()
        wa, wb, wo = kern(blk["a"]), kern(blk["b"]), kern(blk["out"])
        ws = kern(blk["sc"]) if blk["sc"] else None
        with torch.cuda.device(self.dev):
            _C.check(lib.rn_bottleneck64_pack(_C.ptr(wa), _C.ptr(wb), _C.ptr(wo), _C.ptr(ws), self.Cx, _C.ptr(self.packed),
                                              _C.current_stream()), "rn_bottleneck64_pack")
        parts = []
        for op in [blk["a"], blk["b"], blk["out"]] + ([blk["sc"]] if blk["sc"] else []):
            bn = op["bn"]
            f = lambda n: variables[bn + n].to(self.dev, torch.float32)
            scale = f("/gamma") / torch.sqrt(f("/moving_variance") + eps)
            parts += [scale, f("/beta") - f("/moving_mean") * scale]
        self.affine.copy_(torch.cat(parts))
        self._keep = (wa, wb, wo, ws)          # the pack kernel reads them asynchronously

    def launch(self, st):
        _C.check(self.lib.rn_bottleneck64_fwd(self._ref, st), f"rn_bottleneck64_fwd[{self.name}]")
