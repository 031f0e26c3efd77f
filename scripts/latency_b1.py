"""Batch-1 latency of the drop-in calls (host pointers in, host results out): python scripts/latency_b1.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
fa.lib().fh_init(0)
det, rec = fa.FaceDetector(), fa.FaceRecognizer()
assert det.loadModel(models.cached("det_500m_seed100.onnx", models.make_det_500m)) and rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
img = np.random.default_rng(0).integers(0, 256, (640, 640, 3), dtype=np.uint8)
faces = det.detect(img)
f0 = faces[0]
def timeit(f, n=200):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print(f"detect()          640x640 -> {len(faces)} faces: {timeit(lambda: det.detect(img)):.3f} ms")
print(f"extractFeature()  one face                : {timeit(lambda: rec.extractFeature(img, f0)):.3f} ms")
print(f"extractFeatureSimple()                    : {timeit(lambda: rec.extractFeatureSimple(img)):.3f} ms")
