from retinanet.dataloader.anchor_generator import AnchorBoxGenerator
from retinanet.dataloader.input_pipeline import InputPipeline
from retinanet.dataloader.label_encoder import LabelEncoder
from retinanet.dataloader.tfrecord_parser import TFRecordDataset, parse_example
from retinanet.dataloader.utils import normalize_image

__all__ = ["AnchorBoxGenerator", "InputPipeline", "LabelEncoder", "TFRecordDataset", "normalize_image", "parse_example"]
