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
