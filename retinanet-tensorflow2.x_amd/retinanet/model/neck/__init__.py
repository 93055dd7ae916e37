from retinanet.model.neck.builder import build_neck

__all__ = ["build_neck"]
