"""Text-mode counterpart of the reference's demo executable (reference src/main.cpp:264-319):

    python -m facerecognizeonnx_amd.cli detect  <image>            [--det det.onnx]
    python -m facerecognizeonnx_amd.cli compare <image1> <image2>  [--det det.onnx] [--rec rec.onnx]
    python -m facerecognizeonnx_amd.cli simple  <image1> <image2>  [--rec rec.onnx]

Same flows as testDetection / testRecognition / testRecognitionSimple (main.cpp:39-199) — detect, take
faces[0] of each image, extractFeature, compareFaces, threshold 0.6 — but boxes / scores / similarity
are printed instead of drawn (no GUI, no webcam).  Images are read by the library's own cv::imread replacement
(fh_imread: JPEG / PNG / BMP / PPM -> BGR u8) or from `.npy` arrays of shape [rows, cols, 3] (BGR u8).
Model paths default to the reference's (models/det_500m.onnx, models/w600k_r50.onnx, main.cpp:269-270).
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from . import api
from .api import FaceDetector, FaceRecognizer


def imread(path: str):
    if path.endswith(".npy"):
        return np.ascontiguousarray(np.load(path), np.uint8)
    return api.imread(path)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="facerecognizeonnx_amd.cli")
    ap.add_argument("mode", choices=["detect", "compare", "simple"])
    ap.add_argument("images", nargs="+")
    ap.add_argument("--det", default="models/det_500m.onnx")
    ap.add_argument("--rec", default="models/w600k_r50.onnx")
    ap.add_argument("--score", type=float, default=0.5)
    ap.add_argument("--nms", type=float, default=0.4)
    a = ap.parse_args(argv)
    det, rec = FaceDetector(), FaceRecognizer()
    if a.mode != "simple" and not det.loadModel(a.det):
        print("Failed to load face detector model", file=sys.stderr)           # main.cpp:274-278
        return -1
    if a.mode != "detect" and not rec.loadModel(a.rec):
        print("Failed to load face recognizer model", file=sys.stderr)         # main.cpp:280-284
        return -1
    imgs = [imread(p) for p in a.images]
    for p, im in zip(a.images, imgs):
        if im is None:
            print(f"Cannot read image: {p}", file=sys.stderr)                   # main.cpp:43-46
            return -1
    if a.mode == "detect":                                                     # main.cpp:39-65
        faces = det.detect(imgs[0], a.score, a.nms)
        print(f"Detected {len(faces)} faces")
        for i, f in enumerate(faces):
            print(f"Face {i}: box={f.box} score={f.score:.4f} landmarks={np.round(f.landmarks, 1).tolist()}")
        return 0
    if len(imgs) < 2:
        print("need two images", file=sys.stderr)
        return -1
    if a.mode == "compare":                                                    # main.cpp:67-134
        f1s, f2s = det.detect(imgs[0], a.score, a.nms), det.detect(imgs[1], a.score, a.nms)
        print(f"Image 1: {len(f1s)} faces, Image 2: {len(f2s)} faces")
        if not f1s or not f2s:
            print("No face detected in one of the images")
            return -1
        e1, e2 = rec.extractFeature(imgs[0], f1s[0]), rec.extractFeature(imgs[1], f2s[0])
    else:                                                                      # main.cpp:136-199
        e1, e2 = rec.extractFeatureSimple(imgs[0]), rec.extractFeatureSimple(imgs[1])
    if e1.size == 0 or e2.size == 0:
        print("Feature extraction failed")
        return -1
    print(f"Feature dimension: {e1.size}")
    sim = rec.compareFaces(e1, e2)
    print(f"Similarity: {sim:.6f}")
    print("Same person" if sim > 0.6 else "Different person")                  # main.cpp:118
    return 0


if __name__ == "__main__":
    sys.exit(main())
