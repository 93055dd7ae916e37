from retinanet.model.head.builder import build_auxillary_head, build_detection_heads

__all__ = ["build_detection_heads", "build_auxillary_head"]
