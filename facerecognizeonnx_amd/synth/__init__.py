"""Synthetic-model tooling (test / bench infrastructure, not the product path).

The reference ships no model files (reference models/README.md:44-51 says not to
commit them) and none exist offline, so tests and benchmarks build genuine
``.onnx`` files with the public InsightFace architectures
and seeded random weights.  The product loads them through exactly the same
``loadModel(path)`` entry point a real ``det_500m.onnx`` / ``w600k_r50.onnx``
would use.
"""
from .onnx_writer import OnnxBuilder  # noqa: F401
from .models import (  # noqa: F401
    make_iresnet, make_scrfd, make_w600k_r50, make_det_500m, make_predecoded_det,
)
