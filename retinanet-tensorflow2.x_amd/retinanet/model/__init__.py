from retinanet.model.builder import ModelBuilder

__all__ = ["ModelBuilder"]
