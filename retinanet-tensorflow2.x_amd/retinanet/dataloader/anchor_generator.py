"""AnchorBoxGenerator on the GPU (reference retinanet/dataloader/anchor_generator.py:5-112).

Same constructor and properties; `boxes` is a device tensor f32[N,4] of [cx,cy,w,h] rows
produced by rn_anchors_generate (bit-exact against the oracle), `anchor_boundaries` is the
host list of per-level offsets.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from retinanet import _C


class AnchorBoxGenerator:
    def __init__(self, img_h, img_w, min_level, max_level, params, device=None):
        self.image_height = int(img_h)
        self.image_width = int(img_w)
        self.areas = list(params.areas)
        self.aspect_ratios = list(params.aspect_ratios)
        self.scales = list(params.scales)
        self._num_anchors = len(self.aspect_ratios) * len(self.scales)
        self._min_level = int(min_level)
        self._max_level = int(max_level)
        self._strides = [2 ** i for i in range(min_level, max_level + 1)]
        if len(self.areas) != max_level - min_level + 1:
            raise ValueError("anchor_params.areas must have one entry per pyramid level")
        self._device = torch.device(device if device is not None else "cuda")
        self._anchor_boundaries = self._compute_anchor_boundaries()
        self._boxes = self.get_anchors()

    def _compute_anchor_boundaries(self):
        boundaries = [0]
        for i in range(self._min_level, self._max_level + 1):
            n = int(np.ceil(self.image_height / 2 ** i) * np.ceil(self.image_width / 2 ** i) *
                    self._num_anchors)
            boundaries.append(boundaries[-1] + n)
        return boundaries

    def get_anchors(self):
        lib = _C.lib()
        n_total = self._anchor_boundaries[-1]
        boxes = torch.empty((n_total, 4), dtype=torch.float32, device=self._device)
        # the reference divides area / ratio in Python floats before the f32 sqrt (:56)
        aor = [np.float32(a / r) for a in self.areas for r in self.aspect_ratios]
        n_out = ctypes.c_int64(0)
        with torch.cuda.device(self._device):
            _C.check(lib.rn_anchors_generate(
                _C.ptr(boxes), n_total, self.image_height, self.image_width, self._min_level, self._max_level,
                _C.f32_array(self.areas), _C.f32_array(aor), len(self.aspect_ratios),
                _C.f32_array(self.scales), len(self.scales), ctypes.byref(n_out), _C.current_stream()),
                "rn_anchors_generate")
        if n_out.value != n_total:
            raise _C.RnetError(f"anchor count mismatch: {n_out.value} vs {n_total}")
        return boxes

    @property
    def anchor_boundaries(self):
        return self._anchor_boundaries

    @property
    def boxes(self):
        return self._boxes

    @property
    def num_anchors_per_location(self):
        return self._num_anchors
