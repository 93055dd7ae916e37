"""BalanceFeatures as one static graph op — retinanet/model/layers/balance_features.py:19-60: resize every level to
the intermediate level (max-pool below it, nearest up-sampling above), average, resize the average back and add it
to every level (rn_balance_features: one fused launch)."""
from __future__ import annotations

from retinanet.model.graph import Sym


class BalanceFeatures:
    def __init__(self, min_level, max_level, intermediate_level, **_):
        if intermediate_level < min_level or intermediate_level > max_level:
            raise AssertionError("Invalid intermediate level passed")
        self.min_level, self.max_level, self.intermediate_level = int(min_level), int(max_level), int(intermediate_level)

    def __call__(self, features):
        g = next(iter(features.values())).graph
        levels = list(range(self.min_level, self.max_level + 1))
        g.ops.append(dict(op="balance", tensors=[features[str(l)].name for l in levels],
                          mid=self.intermediate_level - self.min_level))
        return {str(l): Sym(g, features[str(l)].name) for l in levels}
