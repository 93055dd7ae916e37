from retinanet.eval.coco_evaluator import COCOEvaluator

__all__ = ["COCOEvaluator"]
