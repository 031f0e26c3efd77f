"""Shared helpers for the test-suite (model builders, seeded inputs)."""
from __future__ import annotations

import os

import numpy as np

from facerecognizeonnx_amd.synth import models

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tiny_iresnet(d, fold_bn=True, seed=1):
    return models.make_iresnet(os.path.join(d, f"r_tiny_{int(fold_bn)}_{seed}.onnx"), (1, 2, 1, 1), (8, 16, 32, 64),
                               112, 512, seed=seed, fold_bn=fold_bn)


def tiny_mbf(d, fold_bn=True, seed=3):
    """MobileFaceNet with 16 base channels: every op kind of the full w600k_mbf graph (grouped 3x3, depthwise + PReLU, GDC, MatMul)."""
    return models.make_mobilefacenet(os.path.join(d, f"m_tiny_{int(fold_bn)}_{seed}.onnx"), (1, 2, 2, 1), 16, 112, 128, seed=seed,
                                     fold_bn=fold_bn)


def tiny_scrfd(d, hw=None, seed=2, cls_bias=-2.0):
    return models.make_scrfd(os.path.join(d, f"s_tiny_{hw}_{seed}.onnx"), (1, 2, 1, 2), (8, 8, 16, 24, 32, 48), 8, 16,
                             seed=seed, cls_bias=cls_bias, static_hw=hw)


def frames_u8(n, rows, cols, seed=0, smooth=False):
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(0, 256, (n, rows, cols, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    out = np.empty((n, rows, cols, 3), np.uint8)
    for i in range(n):
        img = np.zeros((rows, cols, 3), np.float32)
        for c in range(3):
            img[..., c] = 128 + 60 * np.sin(xx * rng.uniform(.02, .2) + rng.uniform(0, 6)) * np.cos(yy * rng.uniform(.02, .2))
        for _ in range(6):
            cx, cy, r = rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(4, min(rows, cols) / 4)
            img += rng.uniform(-80, 80, 3) * np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))[..., None]
        out[i] = np.clip(img + rng.normal(0, 3, img.shape), 0, 255).astype(np.uint8)
    return out


TEMPLATE = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                     [41.5493, 92.3655], [70.7299, 92.2041]], np.float32)


def random_landmarks(n, rows, cols, seed=3, jitter=1.0):
    """Template x random similarity (scale U(1,4), rot U(-30,30) deg, shift inside the frame) + N(0,jitter) px."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 5, 2), np.float32)
    for i in range(n):
        s = rng.uniform(1, 4); th = np.deg2rad(rng.uniform(-30, 30))
        R = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        t = np.array([rng.uniform(0, max(1, cols - 112 * s)), rng.uniform(0, max(1, rows - 112 * s))])
        out[i] = (TEMPLATE @ R.T + t + rng.normal(0, jitter, (5, 2))).astype(np.float32)
    return out


def write_ppm(path, bgr):
    """Binary PPM (P6, RGB order) of a BGR u8 image."""
    h, w, _ = bgr.shape
    with open(path, "wb") as f:
        f.write(f"P6\n{w} {h}\n255\n".encode())
        f.write(np.ascontiguousarray(bgr[:, :, ::-1]).tobytes())


def write_bmp(path, bgr):
    """24-bit bottom-up BMP of a BGR u8 image."""
    import struct
    h, w, _ = bgr.shape
    stride = (w * 3 + 3) & ~3
    rows = b"".join(bgr[y].tobytes() + b"\0" * (stride - w * 3) for y in range(h - 1, -1, -1))
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(rows), 2835, 2835, 0, 0))
        f.write(rows)


def match_records(got, ref, box_tol=1, lm_tol=1e-2, score_tol=1e-4):
    """Pair EVERY record of `ref` (oracle) with one of `got` (GPU): score within `score_tol`, box within `box_tol` px (a 1e-6 network
    difference may flip an int truncation, face_detector.cpp:255-265), landmarks within `lm_tol` px.  Returns the index lists
    (unmatched ref, unmatched got)."""
    used = np.zeros(len(got), bool)
    missing = []
    for i, r in enumerate(ref):
        near = np.where(~used & (np.abs(got["score"] - r["score"]) < score_tol))[0]
        ok = [j for j in near if max(abs(int(got[j][k]) - int(r[k])) for k in ("x", "y", "w", "h")) <= box_tol and
              np.abs(got[j]["lm"] - r["lm"]).max() < lm_tol]
        if ok:
            used[ok[0]] = True
        else:
            missing.append(i)
    return missing, [int(j) for j in np.where(~used)[0]]


def _iou_int(a, b):
    """FaceDetector::iou (face_detector.cpp:340-354): integer intersection, int denominator."""
    x1 = max(int(a["x"]), int(b["x"])); y1 = max(int(a["y"]), int(b["y"]))
    x2 = min(int(a["x"]) + int(a["w"]), int(b["x"]) + int(b["w"])); y2 = min(int(a["y"]) + int(a["h"]), int(b["y"]) + int(b["h"]))
    inter = max(0, x2 - x1) * max(0, y2 - y1)
    den = int(a["w"]) * int(a["h"]) + int(b["w"]) * int(b["h"]) - inter
    return inter / den if den else float("nan")


def assert_records_equivalent(got, ref, score_thr, nms_thr, box_tol=1, lm_tol=1e-2, max_unexplained=0):
    """All records, not a prefix: every oracle record has a GPU counterpart and vice versa.  A record may be unmatched only where a
    ~1e-6 difference of the network outputs can legitimately change the decision: its score lies within 1e-4 of the score threshold
    (strict `>`, face_detector.cpp:253), or its integer IoU with a better-scored record of EITHER list lies within 0.02 of the NMS
    threshold (a +-1 px truncation flip moves the IoU of a small box across the strict `>` of face_detector.cpp:369-371)."""
    missing, surplus = match_records(got, ref, box_tol, lm_tol)
    unexplained = []
    both = list(got) + list(ref)
    for side, idx, arr in (("ref", missing, ref), ("gpu", surplus, got)):
        for i in idx:
            r = arr[i]
            if abs(float(r["score"]) - score_thr) < 1e-4:
                continue
            ious = [_iou_int(r, o) for o in both if float(o["score"]) >= float(r["score"]) - 1e-4]
            if any(abs(v - nms_thr) < 0.02 for v in ious if v == v):
                continue
            unexplained.append((side, i, float(r["score"]), [int(r[k]) for k in ("x", "y", "w", "h")]))
    assert len(unexplained) <= max_unexplained, (len(got), len(ref), unexplained[:8])
    return len(missing), len(surplus)
