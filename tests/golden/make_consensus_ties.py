#!/usr/bin/env python3
"""Generator of tests/golden/consensus_ties.npz: five-point landmark sets on which the 2-point sample models of
cv::estimateAffinePartial2D (reference src/face_recognizer.cpp:110-113; RANSAC, 3 px) reach their LARGEST inlier count (3 or 4 of 5)
with two or more DIFFERENT inlier sets.  OpenCV keeps the first sample that reached the best count (its registrator replaces the best
model only on a strictly larger count), and which sample comes first depends on its fixed-seed RNG order — which this build does not
reproduce: the oracle enumerates the ten pairs in (i < j) order.  These cases pin the build's documented rule ("most inliers, then the
first pair") on both the oracle and the GPU, and scripts/make_reference_goldens.py emits OpenCV's own answer for exactly these
landmarks when a user has cv2, which settles whether the two orders ever disagree in practice.

Pure numpy, seeded; reads nothing outside this file.  Run: python tests/golden/make_consensus_ties.py
"""
import os

import numpy as np

TEMPLATE = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]], np.float32)
PAIRS = [(i, j) for i in range(5) for j in range(i + 1, 5)]


def pair_masks(lm):
    """Inlier mask (bit p = point p) of each of the ten 2-point similarity models lm -> TEMPLATE, fp64, squared error <= 9."""
    f = lm.astype(np.float64); t = TEMPLATE.astype(np.float64)
    out = []
    for i, j in PAIRS:
        d = f[i] - f[j]; den = d @ d
        if not den > 0:
            out.append(0); continue
        D = t[i] - t[j]
        a = (D[0] * d[0] + D[1] * d[1]) / den; b = (D[1] * d[0] - D[0] * d[1]) / den
        tx = t[i, 0] - (a * f[i, 0] - b * f[i, 1]); ty = t[i, 1] - (b * f[i, 0] + a * f[i, 1])
        ex = a * f[:, 0] - b * f[:, 1] + tx - t[:, 0]; ey = b * f[:, 0] + a * f[:, 1] + ty - t[:, 1]
        e = ex * ex + ey * ey
        out.append(int(sum(1 << p for p in range(5) if e[p] <= 9.0)))
    return out


def margin(lm):
    """Smallest distance of any (pair model, point) squared error from the threshold 9: cases too close to it are rejected, so that
    fp32 landmark rounding can never move a point across the threshold."""
    f = lm.astype(np.float64); t = TEMPLATE.astype(np.float64)
    m = np.inf
    for i, j in PAIRS:
        d = f[i] - f[j]; den = d @ d
        D = t[i] - t[j]
        a = (D[0] * d[0] + D[1] * d[1]) / den; b = (D[1] * d[0] - D[0] * d[1]) / den
        tx = t[i, 0] - (a * f[i, 0] - b * f[i, 1]); ty = t[i, 1] - (b * f[i, 0] + a * f[i, 1])
        ex = a * f[:, 0] - b * f[:, 1] + tx - t[:, 0]; ey = b * f[:, 0] + a * f[:, 1] + ty - t[:, 1]
        m = min(m, np.abs(ex * ex + ey * ey - 9.0).min())
    return m


def main():
    rng = np.random.default_rng(20261004)
    want = {3: 8, 4: 8}
    cases, counts = [], []
    while any(v > 0 for v in want.values()):
        s = rng.uniform(1, 4); th = np.deg2rad(rng.uniform(-30, 30))
        R = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        t = np.array([rng.uniform(0, 640 - 112 * s), rng.uniform(0, 480 - 112 * s)])
        off = rng.normal(0, rng.uniform(1.0, 6.0), (5, 2))                    # residuals around the 3 px threshold, in template pixels
        lm = ((TEMPLATE + off) @ R.T + t).astype(np.float32)
        masks = pair_masks(lm)
        pc = [bin(m).count("1") for m in masks]
        best = max(pc)
        sets = {m for m, c in zip(masks, pc) if c == best}
        if best in want and want[best] > 0 and len(sets) >= 2 and margin(lm) > 1e-2:
            want[best] -= 1
            cases.append(lm); counts.append(best)
    order = np.argsort(counts, kind="stable")
    lm = np.stack(cases)[order]; counts = np.array(counts)[order]
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "consensus_ties.npz"), landmarks=lm, best_count=counts)
    print("wrote consensus_ties.npz:", lm.shape, counts.tolist())


if __name__ == "__main__":
    main()
