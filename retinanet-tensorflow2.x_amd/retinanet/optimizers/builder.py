"""Learning-rate schedules and the optimizer config of the reference
(retinanet/optimizers/builder.py:13-71, cosine_decay_with_warmup.py:4-43,
piecewise_constant_decay_with_warmup.py:4-35).

The schedules are host-side scalar functions of the step; the optimizer object only carries
the hyper-parameters the fused HIP step kernel consumes (SGD momentum, EMA decay, clipnorm).
"""
from __future__ import annotations

import math
from copy import deepcopy


class CosineDecayWithLinearWarmup:
    def __init__(self, initial_learning_rate, warmup_learning_rate, warmup_steps, total_steps, alpha=0.0):
        self.initial_learning_rate = float(initial_learning_rate)
        self.warmup_learning_rate = float(warmup_learning_rate)
        self.warmup_steps = int(warmup_steps)
        self.decay_steps = int(total_steps) - int(warmup_steps)
        self.alpha = float(alpha)
        self._step_size = self.initial_learning_rate - self.warmup_learning_rate

    def __call__(self, step):
        if step < self.warmup_steps:
            return self.warmup_learning_rate + step / self.warmup_steps * self._step_size
        # Keras CosineDecay evaluated on the un-shifted step (cosine_decay_with_warmup.py:14,32-33)
        s = min(step, self.decay_steps)
        cosine = 0.5 * (1.0 + math.cos(math.pi * s / self.decay_steps))
        return self.initial_learning_rate * ((1.0 - self.alpha) * cosine + self.alpha)


class PiecewiseConstantDecayWithLinearWarmup:
    def __init__(self, warmup_learning_rate, warmup_steps, boundaries, values):
        self.boundaries = [b - 1 for b in boundaries]  # piecewise_constant_decay_with_warmup.py:8-9
        self.values = list(values)
        self.warmup_learning_rate = float(warmup_learning_rate)
        self.warmup_steps = int(warmup_steps)
        self._step_size = self.values[0] - self.warmup_learning_rate

    def __call__(self, step):
        if step < self.warmup_steps:
            return self.warmup_learning_rate + step / self.warmup_steps * self._step_size
        for b, v in zip(self.boundaries, self.values):
            if step <= b:
                return v
        return self.values[-1]


def get_learning_rate_schedule(total_steps, params):
    _params = dict(deepcopy(params))
    schedule_type = _params.pop("schedule_type", None)
    if schedule_type == "piecewise_constant_decay":
        return PiecewiseConstantDecayWithLinearWarmup(**_params)
    if schedule_type == "cosine_decay":
        _params["total_steps"] = total_steps
        return CosineDecayWithLinearWarmup(**_params)
    raise ValueError("Invalid learning rate schedule requested")


class OptimizerConfig:
    """What `build_optimizer` returns: hyper-parameters + schedule + step counter."""

    def __init__(self, name, momentum, nesterov, clipnorm, learning_rate, use_moving_average,
                 moving_average_decay, loss_scale):
        self.name = name
        self.momentum = momentum
        self.nesterov = nesterov
        self.clipnorm = clipnorm
        self.learning_rate = learning_rate
        self.use_moving_average = use_moving_average
        self.moving_average_decay = moving_average_decay
        # tf.keras.mixed_precision.LossScaleOptimizer(dynamic=True) defaults (optimizers/builder.py:56-64)
        self.dynamic_loss_scale = loss_scale
        self.initial_loss_scale = 2.0 ** 15
        self.loss_scale_growth_steps = 2000
        self.iterations = 0

    def lr(self, step=None):
        return self.learning_rate(self.iterations if step is None else step)

    def ema_decay(self, step=None):
        """tfa MovingAverage(dynamic_decay=True): min(decay, (1+t)/(10+t))."""
        t = self.iterations if step is None else step
        return min(self.moving_average_decay, (1.0 + t) / (10.0 + t))


def build_optimizer(params, train_steps, precision):
    _params = dict(deepcopy(params))
    lr_params = _params.pop("lr_params", None)
    use_moving_average = bool(_params.pop("use_moving_average", False))
    moving_average_decay = _params.pop("moving_average_decay", 0.0) or 0.0
    _params.pop("global_clipnorm", None)
    clipnorm = _params.pop("clipnorm", None)
    name = str(_params.pop("name", "sgd")).lower()
    if name != "sgd":
        raise ValueError(f"optimizer {name}: every shipped config trains with SGD momentum")
    return OptimizerConfig(name=name, momentum=float(_params.get("momentum", 0.0)),
                           nesterov=bool(_params.get("nesterov", False)), clipnorm=clipnorm,
                           learning_rate=get_learning_rate_schedule(train_steps, lr_params),
                           use_moving_average=use_moving_average,
                           moving_average_decay=float(moving_average_decay),
                           loss_scale=(precision == "mixed_float16"))
