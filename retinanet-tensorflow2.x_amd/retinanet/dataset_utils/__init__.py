from retinanet.dataset_utils.tfrecord_writer import TFrecordWriter

__all__ = ["TFrecordWriter"]
