"""Consumes TRUE-REFERENCE goldens when a user has produced them with scripts/make_reference_goldens.py (ONNX Runtime / OpenCV on the
genuine model files, in the build container); skipped otherwise.  These are the only fixtures that can pin the oracle to the
reference's own numeric engines: absent them, parity stays "unpinned" (DESIGN.md §0).

ref_opencv.npz needs no model file.  ref_det.npz / ref_rec.npz also need FACEHIP_REF_DET / FACEHIP_REF_REC to point at the model files
the goldens were made from (they are never committed).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import onnx_min, oracle
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    p = os.path.join(util.GOLDEN, name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not present: run scripts/make_reference_goldens.py where ONNX Runtime / OpenCV and the genuine models exist")
    return np.load(p)


def test_generator_is_inert_by_default(tmp_path):
    """No arguments -> nothing is written, exit code 0 (it must never fetch or assume anything)."""
    before = set(os.listdir(util.GOLDEN))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_reference_goldens.py")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "inert by design" in p.stdout
    assert set(os.listdir(util.GOLDEN)) == before
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_reference_goldens.py"), "--det", str(tmp_path / "nope.onnx")],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "nothing written" in p.stdout
    assert set(os.listdir(util.GOLDEN)) == before


def test_oracle_resize_and_warp_match_opencv():
    g = _load("ref_opencv.npz")
    for i in range(4):
        src, dst = g[f"resize{i}_src"], g[f"resize{i}_dst"]
        got = oracle.resize_bilinear(src, dst.shape[1], dst.shape[0])
        assert np.array_equal(got, dst), (i, int(np.abs(got.astype(int) - dst.astype(int)).max()))
    img = g["warp_img"]
    for lm, M, crop, ok in zip(g["warp_lm"], g["warp_M"], g["warp_crop"], g["warp_ok"]):
        mine = oracle.estimate_similarity(lm.reshape(-1))
        assert (mine is not None) == bool(ok)
        if not ok:
            continue
        np.testing.assert_allclose(mine, M, rtol=1e-6, atol=1e-6)          # RANSAC-equivalent consensus + LS refit vs OpenCV's own
        assert np.array_equal(oracle.warp_affine(img, M), crop)            # the fixed-point bilinear path, given OpenCV's matrix


def test_oracle_consensus_ties_match_opencv():
    """The one place where the restatement's pair order ("most inliers, then the first pair in (i < j) order") and OpenCV's fixed-seed
    RANSAC sample order can disagree: equal-count ties between different inlier sets (tests/golden/consensus_ties.npz).  With cv2's
    own answers present, every case is compared; a disagreement is reported with the two inlier sets, not hidden."""
    g = _load("ref_opencv.npz")
    if "tie_lm" not in g.files:
        pytest.skip("ref_opencv.npz predates the tie cases: re-run scripts/make_reference_goldens.py --opencv")
    bad = []
    for i, (lm, M, ok) in enumerate(zip(g["tie_lm"], g["tie_M"], g["tie_ok"])):
        mine = oracle.estimate_similarity(lm.reshape(-1))
        if (mine is not None) != bool(ok) or (ok and not np.allclose(mine, M, rtol=1e-6, atol=1e-6)):
            bad.append((i, g["tie_inliers"][i].tolist()))
    assert not bad, f"OpenCV chose another consensus set than the first-pair rule on cases {bad}"


def test_oracle_det_network_matches_onnxruntime():
    g = _load("ref_det.npz")
    path = os.environ.get("FACEHIP_REF_DET")
    if not path or not os.path.isfile(path):
        pytest.skip("FACEHIP_REF_DET does not point at the model file the goldens were made from")
    od = oracle.OracleDetector()
    assert od.loadModel(path)
    frames = g["frames"]
    for i in range(len(frames)):
        inp, scale = oracle.det_preprocess(frames[i], od.inW, od.inH)
        outs = od.run_network(inp)
        for j, o in enumerate(outs):
            ref = g[f"f{i}_o{j}"]
            np.testing.assert_allclose(np.asarray(o).reshape(ref.shape), ref, rtol=1e-4, atol=1e-4)


def test_oracle_rec_network_matches_onnxruntime():
    g = _load("ref_rec.npz")
    path = os.environ.get("FACEHIP_REF_REC")
    if not path or not os.path.isfile(path):
        pytest.skip("FACEHIP_REF_REC does not point at the model file the goldens were made from")
    orc = oracle.OracleRecognizer()
    assert orc.loadModel(path)
    for crop, feat in zip(g["crops"], g["features"]):
        r = oracle.run_graph(orc.g, {orc.g.inputs[0][0]: oracle.rec_preprocess(crop)[None]})[orc.g.outputs[0][0]].reshape(-1)
        assert np.abs(r - feat).max() < 1e-4 * max(1.0, np.abs(feat).max())
        a, b = oracle.l2_normalize(r.copy()), oracle.l2_normalize(feat.copy())
        assert 1.0 - float(np.dot(a, b)) < 1e-5
