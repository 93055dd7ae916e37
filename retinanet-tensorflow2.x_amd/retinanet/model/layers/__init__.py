from retinanet.model.layers.postprocessing_ops import (DetectionPostProcess, FilterTopKDetections,
                                                       FuseDetections, GenerateDetections,
                                                       TransformBoxesAndScores)

__all__ = ["DetectionPostProcess", "FilterTopKDetections", "FuseDetections", "GenerateDetections",
           "TransformBoxesAndScores"]
