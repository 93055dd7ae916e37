from retinanet.cfg.config import AttrDict, Config, default_params, efficientnet_params

__all__ = ["AttrDict", "Config", "default_params", "efficientnet_params"]
