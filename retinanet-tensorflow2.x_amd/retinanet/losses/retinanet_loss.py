"""RetinaNetLoss on the GPU (reference retinanet/losses/retinanet_loss.py:9-83 and
retinanet/losses/loss_impl.py:4-105).

`RetinaNetLoss(num_classes, params)(targets, predictions)` returns the reference's loss dict.
One fused HIP launch set computes the focal + Huber sums AND the gradients with respect to
every level's logits / box predictions (stored in `.grads` for the backward pass), so the
autograd tape of the reference is replaced by closed-form derivatives.

Distributed: the normaliser is all-reduced over `process_group` exactly where the reference
calls replica_context.all_reduce (retinanet_loss.py:46-49).
"""
from __future__ import annotations

import torch

from retinanet import _C


class RetinaNetLoss:
    def __init__(self, num_classes, params, process_group=None):
        self._num_classes = int(num_classes)
        self._alpha = float(params.focal_loss.alpha)
        self._gamma = float(params.focal_loss.gamma)
        self._label_smoothing = float(params.focal_loss.label_smoothing)
        self._delta = float(params.smooth_l1_loss.delta)
        self._box_loss_weight = float(params.box_loss_weight)
        self._class_loss_weight = float(params.class_loss_weight)
        self._auxillary_loss_weight = float(params.auxillary_loss_weight)
        if params.normalizer.use_moving_average:
            raise NotImplementedError("loss.normalizer.use_moving_average is false in every shipped config "
                                      "and is out of scope (SURVEY §2.2 C6)")
        self._pg = process_group
        self._ws = None
        self.grads = None

    def _num_replicas(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self._pg)
        return 1

    def __call__(self, targets, predictions, compute_grads=True, grad_scale=None, grads_bf16=None, normalizer=None):
        """normalizer: optional device f32[1] = all_reduce_sum(sum(num-positives) + 1) / replicas already computed by
        the caller (the training engine folds that scalar into its first SyncBN message) — else computed here.
        grads_bf16 = {"class-predictions": {level: bf16[B,H,W,stride]}, "box-predictions": {...}}: write the
        gradients as bf16 into these (channel-padded) tensors instead of fp32 `self.grads` — the training engine
        passes the dy tensors of the prediction convs, so no fp32 gradient is materialised."""
        lib = _C.lib()
        cls_pred = predictions["class-predictions"]
        box_pred = predictions["box-predictions"]
        levels = sorted(cls_pred.keys(), key=int)
        flat = targets["_flat"]
        cls_t, box_t = flat["class-targets"], flat["box-targets"]
        dev = cls_t.device
        B = cls_t.shape[0]
        K = self._num_classes
        # normaliser = all_reduce_sum(sum(num-positives) + 1) / replicas  (retinanet_loss.py:38-49)
        from retinanet.distribute import global_normalizer
        R = self._num_replicas()
        if normalizer is None:
            normalizer = global_normalizer(targets["num-positives"].sum(), R, self._pg)
        else:
            normalizer = normalizer.reshape(1).to(torch.float32).contiguous()
        offs = [0]
        cl, bl = [], []
        for lv in levels:
            c = cls_pred[lv]
            b = box_pred[lv]
            if c.dtype != torch.float32 or b.dtype != torch.float32:
                raise TypeError("predictions must be float32 (the reference's prediction convs are fp32)")
            c = c.contiguous()
            b = b.contiguous()
            n_l = c.numel() // (B * K)
            offs.append(offs[-1] + n_l)
            cl.append(c)
            bl.append(b)
        if offs[-1] != cls_t.shape[1]:
            raise ValueError(f"predictions cover {offs[-1]} anchors, targets {cls_t.shape[1]}")
        bf16_out = compute_grads and grads_bf16 is not None
        dcl = [torch.empty_like(c) for c in cl] if compute_grads and not bf16_out else None
        dbl = [torch.empty_like(b) for b in bl] if compute_grads and not bf16_out else None
        need = lib.rn_loss_workspace_bytes(B, offs[-1], K)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        out = torch.empty((4,), dtype=torch.float32, device=dev)
        if grad_scale is None:
            grad_scale = 1.0 / R  # per_replica_loss = total / replicas  (executor.py:421)
        with torch.cuda.device(dev):
            if bf16_out:
                gc = [grads_bf16["class-predictions"][lv] for lv in levels]
                gb = [grads_bf16["box-predictions"][lv] for lv in levels]
                na = cl[0].shape[-1] // K
                for c, b, tc, tb in zip(cl, bl, gc, gb):
                    if (tc.dtype not in (torch.bfloat16, torch.float16) or tb.dtype != tc.dtype or not tc.is_contiguous()
                            or not tb.is_contiguous() or tc.shape[:-1] != c.shape[:-1] or tb.shape[:-1] != b.shape[:-1]
                            or tc.shape[-1] != gc[0].shape[-1] or tb.shape[-1] != gb[0].shape[-1]):
                        raise ValueError("grads_bf16 tensors must be contiguous 16-bit [B,H,W,stride] like the predictions")
                lib = _C.lib(gc[0].dtype == torch.float16)   # the build whose 16-bit type the gradient tensors hold
                _C.check(lib.rn_retinanet_loss_fwd_bwd_bf16(
                    _C.ptr_array(cl), _C.ptr_array(bl), _C.ptr_array(gc), _C.ptr_array(gb), gc[0].shape[-1],
                    gb[0].shape[-1], na, _C.i64_array(offs), len(levels), B, K, _C.ptr(cls_t), _C.ptr(box_t),
                    _C.ptr(normalizer), self._alpha, self._gamma, self._label_smoothing, self._delta,
                    self._box_loss_weight, self._class_loss_weight, float(grad_scale), _C.ptr(out), _C.ptr(self._ws),
                    self._ws.numel(), _C.current_stream()), "rn_retinanet_loss_fwd_bwd_bf16")
            else:
                _C.check(lib.rn_retinanet_loss_fwd_bwd(
                    _C.ptr_array(cl), _C.ptr_array(bl), _C.ptr_array(dcl), _C.ptr_array(dbl),
                    _C.i64_array(offs), len(levels), B, K, _C.ptr(cls_t), _C.ptr(box_t), _C.ptr(normalizer),
                    self._alpha, self._gamma, self._label_smoothing, self._delta, self._box_loss_weight,
                    self._class_loss_weight, float(grad_scale), _C.ptr(out), _C.ptr(self._ws), self._ws.numel(),
                    _C.current_stream()), "rn_retinanet_loss_fwd_bwd")
        if bf16_out:
            self.grads = None
        elif compute_grads:
            self.grads = {"class-predictions": dict(zip(levels, dcl)), "box-predictions": dict(zip(levels, dbl))}
        return {"box-loss": out[0], "class-loss": out[1], "weighted-loss": out[2],
                "num-anchors-matched": out[3], "iou-prediction-loss": 0.0}
