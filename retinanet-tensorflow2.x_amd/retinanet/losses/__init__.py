from retinanet.losses.retinanet_loss import RetinaNetLoss

__all__ = ["RetinaNetLoss"]
