from retinanet.model.backbone.builder import build_backbone

__all__ = ["build_backbone"]
