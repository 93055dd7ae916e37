from retinanet.dataloader.anchor_generator import AnchorBoxGenerator
from retinanet.dataloader.label_encoder import LabelEncoder

__all__ = ["AnchorBoxGenerator", "LabelEncoder"]
