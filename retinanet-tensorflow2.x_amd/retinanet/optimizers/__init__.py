from retinanet.optimizers.builder import build_optimizer, get_learning_rate_schedule

__all__ = ["build_optimizer", "get_learning_rate_schedule"]
