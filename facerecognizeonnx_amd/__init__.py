"""facerecognizeonnx_amd — MI355X-native detect -> align -> embed -> compare.

Drop-in for the hot path of cucibala/FaceRecognizeOnnx (FaceDetector::detect,
FaceRecognizer::extractFeature / compareFaces) behind a C ABI (include/facehip.h) implemented
as hand-written gfx950 HIP kernels (csrc/).  See DESIGN.md.
"""
from .api import (Comm, FaceBox, FaceDetector, FaceRecognizer, FrameStream, Gallery, pipeline_run_dev, pipeline_submit_dev,  # noqa: F401
                  imread, plan_describe)
from ._lib import FACE_DTYPE, FaceHipError, build, lib  # noqa: F401

__all__ = ["Comm", "FaceBox", "FaceDetector", "FaceRecognizer", "Gallery", "FrameStream", "pipeline_run_dev", "pipeline_submit_dev", "plan_describe", "imread",
           "FACE_DTYPE", "FaceHipError", "build", "lib"]
