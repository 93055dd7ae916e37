"""Activation / normalisation selection of the reference (retinanet/model/utils.py:7-37) for the static layer graph.

The reference returns Keras layer factories; here a layer is an entry of the static graph, so the "op" is what the
graph needs to know about it: the activation's name (the conv / BatchNorm kernels apply it in their epilogue) and,
for BatchNorm, whether the statistics are all-reduced over the replicas (SyncBatchNormalization iff `use_sync` and
more than one replica is in sync — model/utils.py:10-12) together with momentum / epsilon."""
from __future__ import annotations

_ACTIVATIONS = ("relu", "relu6", "swish")


def get_activation_op(activation_type):
    """model/utils.py:25-37: 'relu' | 'relu6' | 'swish' (anything else raises like the reference)."""
    if activation_type not in _ACTIVATIONS:
        raise ValueError("{} activation not implemented".format(activation_type))
    return activation_type


def get_normalization_op(use_sync=False, num_replicas=1, momentum=0.99, epsilon=0.001, **_):
    """model/utils.py:7-22.  -> dict(kind, sync, momentum, epsilon): `sync` selects the all-reduced statistics of
    tf.keras.layers.experimental.SyncBatchNormalization, taken only when more than one replica takes part."""
    sync = bool(use_sync) and int(num_replicas) > 1
    return {"kind": "sync_batch_normalization" if sync else "batch_normalization", "sync": sync,
            "momentum": float(momentum), "epsilon": float(epsilon)}
