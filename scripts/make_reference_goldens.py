#!/usr/bin/env python3
"""Optional TRUE-REFERENCE golden vectors — the only route by which the oracle's parity status can ever leave "unpinned".

The reference delegates its arithmetic to ONNX Runtime and OpenCV (reference src/face_detector.cpp:24-26,117,125,179-183;
src/face_recognizer.cpp:110-113,130,138,279-283) and ships neither golden vectors nor the two model files
(models/README.md:44-51).  Nothing of that exists in the build image, so by default this script does NOTHING and exits 0 with a
message.  It becomes active only when the person running it supplies, in the BUILD container,

    --det /path/to/det_500m.onnx   and / or   --rec /path/to/w600k_r50.onnx        (the genuine InsightFace files)
    an importable `onnxruntime` and / or `cv2`

It never downloads anything, never imports or reads anything under /root/reference, and ships no reference code: it calls the same
two third-party libraries the reference calls, with the reference's own call parameters, on seeded inputs, and writes

    tests/golden/ref_opencv.npz   cv2.resize (INTER_LINEAR), cv2.warpAffine, cv2.estimateAffinePartial2D on seeded inputs and on the
                                  committed equal-count tie cases (tests/golden/consensus_ties.npz)
                                  (needs cv2 only — pins the fixed-point restatements of SURVEY.md App. B)
    tests/golden/ref_det.npz      ORT CPU outputs of det_500m on seeded 640x640 frames preprocessed as face_detector.cpp:92-137
    tests/golden/ref_rec.npz      ORT CPU outputs of w600k_r50 on seeded 112x112 crops preprocessed as face_recognizer.cpp:135-150

`tests/test_reference_goldens.py` consumes whichever of these files exist (the ORT ones also need FACEHIP_REF_DET / FACEHIP_REF_REC to
point at the same model files, which are never committed) and is skipped otherwise.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
TEMPLATE = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655], [70.7299, 92.2041]], np.float32)


def _try(name):
    try:
        return __import__(name)
    except Exception as e:  # noqa: BLE001 - any import failure means "not available here"
        print(f"[make_reference_goldens] {name} is not importable here ({type(e).__name__}): skipping what needs it")
        return None


def _frames(n, rows, cols, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    out = np.empty((n, rows, cols, 3), np.uint8)
    for i in range(n):
        img = np.zeros((rows, cols, 3), np.float32)
        for c in range(3):
            img[..., c] = 128 + 60 * np.sin(xx * rng.uniform(.02, .2) + rng.uniform(0, 6)) * np.cos(yy * rng.uniform(.02, .2))
        img += rng.normal(0, 12, img.shape)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def opencv_goldens(cv2):
    rng = np.random.default_rng(1234)
    out = {"cv2_version": np.array(cv2.__version__)}
    # cv::resize(src, dst, Size(w, h)) — face_detector.cpp:117, face_recognizer.cpp:123,170 (default INTER_LINEAR)
    srcs = [_frames(1, 97, 131, 1)[0], _frames(1, 480, 640, 2)[0], _frames(1, 224, 224, 3)[0], _frames(1, 33, 20, 4)[0]]
    dsts = [(112, 112), (640, 480), (112, 112), (112, 112)]
    for i, (s, (dw, dh)) in enumerate(zip(srcs, dsts)):
        out[f"resize{i}_src"] = s
        out[f"resize{i}_dst"] = cv2.resize(s, (dw, dh))
    # estimateAffinePartial2D(landmarks -> template) + warpAffine(image, M, 112x112) — face_recognizer.cpp:110-113,129-130
    img = _frames(1, 480, 640, 5)[0]
    out["warp_img"] = img
    lms, Ms, crops, oks = [], [], [], []
    for i in range(24):
        s = rng.uniform(1, 4); th = np.deg2rad(rng.uniform(-30, 30))
        R = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        t = np.array([rng.uniform(0, 640 - 112 * s), rng.uniform(0, 480 - 112 * s)])
        lm = (TEMPLATE @ R.T + t + rng.normal(0, 1.0 if i % 3 else 0.0, (5, 2))).astype(np.float32)
        if i % 8 == 7:
            lm[rng.integers(0, 5)] += 25.0                                 # one outlier landmark: RANSAC must drop it
        M, _ = cv2.estimateAffinePartial2D(lm, TEMPLATE)
        lms.append(lm)
        oks.append(M is not None)
        Ms.append(M if M is not None else np.zeros((2, 3)))
        crops.append(cv2.warpAffine(img, M, (112, 112)) if M is not None else np.zeros((112, 112, 3), np.uint8))
    # equal-count consensus ties between DIFFERENT inlier sets (tests/golden/consensus_ties.npz, generator beside it): the one place
    # where this build's "most inliers, then the first pair in (i < j) order" and OpenCV's fixed-seed sample order may disagree.
    # cv2's own M / inlier mask for exactly those landmarks settles it (tests/test_reference_goldens.py compares).
    ties = os.path.join(GOLDEN, "consensus_ties.npz")
    if os.path.isfile(ties):
        tl = np.load(ties)["landmarks"]
        tM, tin, tok = [], [], []
        for lm in tl:
            M, inl = cv2.estimateAffinePartial2D(lm, TEMPLATE)
            tok.append(M is not None)
            tM.append(M if M is not None else np.zeros((2, 3)))
            tin.append(np.asarray(inl).reshape(-1) if inl is not None else np.zeros(5, np.uint8))
        out["tie_lm"] = tl; out["tie_M"] = np.stack(tM); out["tie_inliers"] = np.stack(tin); out["tie_ok"] = np.array(tok)
    out["warp_lm"] = np.stack(lms); out["warp_M"] = np.stack(Ms); out["warp_crop"] = np.stack(crops); out["warp_ok"] = np.array(oks)
    np.savez_compressed(os.path.join(GOLDEN, "ref_opencv.npz"), **out)
    print(f"[make_reference_goldens] wrote tests/golden/ref_opencv.npz (OpenCV {cv2.__version__})")


def _session(ort, path):
    so = ort.SessionOptions()
    so.intra_op_num_threads = 4                                            # face_detector.cpp:10, face_recognizer.cpp:11
    so.graph_optimization_level = ort.GraphOptimizationLevel.ORT_ENABLE_ALL   # :11 / :12
    return ort.InferenceSession(path, so, providers=["CPUExecutionProvider"])


def det_goldens(ort, path):
    sess = _session(ort, path)
    frames = _frames(3, 640, 640, 11)
    inp = ((frames[..., ::-1].astype(np.float32) - 127.5) / 128.0).transpose(0, 3, 1, 2).copy()      # face_detector.cpp:125-136
    name = sess.get_inputs()[0].name
    outs = [sess.run(None, {name: inp[i:i + 1]}) for i in range(len(frames))]                       # batch 1, all outputs (:170-183)
    d = {"ort_version": np.array(ort.__version__), "frames": frames, "output_names": np.array([o.name for o in sess.get_outputs()])}
    for i, o in enumerate(outs):
        for j, t in enumerate(o):
            d[f"f{i}_o{j}"] = t
    np.savez_compressed(os.path.join(GOLDEN, "ref_det.npz"), **d)
    print(f"[make_reference_goldens] wrote tests/golden/ref_det.npz (ONNX Runtime {ort.__version__}, {len(outs[0])} outputs per frame)")


def rec_goldens(ort, path):
    sess = _session(ort, path)
    crops = _frames(6, 112, 112, 12)
    inp = ((crops[..., ::-1].astype(np.float32) - 127.5) / 128.0).transpose(0, 3, 1, 2).copy()       # face_recognizer.cpp:138-149
    name = sess.get_inputs()[0].name
    feats = np.stack([sess.run(None, {name: inp[i:i + 1]})[0].reshape(-1) for i in range(len(crops))])  # :270-294
    np.savez_compressed(os.path.join(GOLDEN, "ref_rec.npz"), ort_version=np.array(ort.__version__), crops=crops, features=feats)
    print(f"[make_reference_goldens] wrote tests/golden/ref_rec.npz (ONNX Runtime {ort.__version__}, feature dim {feats.shape[1]})")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--det", help="path to the genuine det_500m.onnx (user-supplied; never fetched)")
    ap.add_argument("--rec", help="path to the genuine w600k_r50.onnx (user-supplied; never fetched)")
    ap.add_argument("--opencv", action="store_true", help="write the OpenCV-only goldens (needs cv2)")
    a = ap.parse_args(argv)
    if not (a.det or a.rec or a.opencv):
        print("[make_reference_goldens] nothing requested (no --det / --rec / --opencv): inert by design, nothing written.")
        return 0
    wrote = False
    if a.opencv:
        cv2 = _try("cv2")
        if cv2 is not None:
            opencv_goldens(cv2); wrote = True
    if a.det or a.rec:
        ort = _try("onnxruntime")
        for path, fn in ((a.det, det_goldens), (a.rec, rec_goldens)):
            if not path:
                continue
            if not os.path.isfile(path):
                print(f"[make_reference_goldens] {path}: no such file (this script never downloads models)")
                continue
            if ort is not None:
                fn(ort, path); wrote = True
    if not wrote:
        print("[make_reference_goldens] nothing written.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
