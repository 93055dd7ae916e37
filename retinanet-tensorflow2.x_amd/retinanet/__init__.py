"""MI355X-native RetinaNet hot path behind the `retinanet` Python surface of
srihari-humbarwadi/retinanet-tensorflow2.x (config JSON, ModelBuilder, losses, anchors,
post-processing).  Host code is Python; all device work is hand-written HIP for gfx950 in
librnet_hip.so, loaded through ctypes (`retinanet._C`).  PyTorch is used for device memory,
streams and torch.distributed (RCCL) only.
"""
__version__ = "0.1.0"


def __getattr__(name):
    # `from retinanet import Executor` (reference retinanet/__init__.py) — imported lazily so that
    # `import retinanet` does not pull torch / the HIP library in for the light-weight cfg / _C users
    if name == "Executor":
        from retinanet.executor import Executor
        return Executor
    raise AttributeError(name)


__all__ = ["Executor"]
