#!/opt/conda/bin/python3.9
"""Golden vectors for the 5-point similarity estimate from an INDEPENDENT implementation: scikit-image 0.18.3
(`skimage.transform.SimilarityTransform.estimate`, Umeyama's closed form) in the build container's conda tree
(/opt/conda, python 3.9 — the only place skimage exists; it is not importable from the test interpreter).

    /opt/conda/bin/python3.9 tests/golden/make_similarity_golden.py      ->  tests/golden/similarity_skimage.npz

What it pins: the least-squares refit at the end of `orc_estimate_similarity5` (oracle/face_oracle.c) and of the GPU
`estimate_similarity5` — i.e. what cv::estimateAffinePartial2D returns when all five correspondences are inliers
(reference src/face_recognizer.cpp:110-113): a rotation + uniform scale + translation minimising the squared
residuals.  It does NOT pin OpenCV's RANSAC sampling (cases with outliers are not in the file): parity with OpenCV
itself stays unpinned, this narrows the risk on the arithmetic both restatements share.
"""
import os

import numpy as np
from skimage.transform import SimilarityTransform

TEMPLATE = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]], np.float32)

rng = np.random.default_rng(20260410)
src, exp = [], []
for i in range(96):
    s = rng.uniform(0.8, 5.0)
    th = np.deg2rad(rng.uniform(-75, 75))
    R = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    t = rng.uniform(-50, 900, 2)
    jitter = 0.0 if i < 8 else rng.uniform(0.05, 0.6) * s          # well inside the 3 px consensus threshold after un-scaling
    pts = (TEMPLATE.astype(np.float64) @ R.T + t + rng.normal(0, jitter, (5, 2))).astype(np.float32)
    tf = SimilarityTransform()
    assert tf.estimate(pts.astype(np.float64), TEMPLATE.astype(np.float64))
    resid = np.linalg.norm(tf(pts.astype(np.float64)) - TEMPLATE, axis=1)
    assert resid.max() < 2.0, resid                                  # every point an inlier for any hypothesis near the optimum
    src.append(pts)
    exp.append(tf.params[:2].copy())
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "similarity_skimage.npz")
np.savez_compressed(out, src=np.stack(src), expected=np.stack(exp), template=TEMPLATE, skimage_version=np.array("0.18.3"))
print("wrote", out, np.stack(src).shape)
