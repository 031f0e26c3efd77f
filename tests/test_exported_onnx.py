"""Ingest of ONNX files written by a FOREIGN exporter (torch.onnx.export, TorchScript exporter, opset 11) — fixtures and their
generator: tests/golden/make_exported_golden.py.

The reference hands any valid graph to ONNX Runtime (`new Ort::Session(env, path, opts)`, reference src/face_detector.cpp:24-26,
src/face_recognizer.cpp:25-27); every other model file in this repo is written by the build's own `synth/onnx_writer.py`.  These graphs
carry the habits of a real exporter: `Identity(initializer)` aliases of de-duplicated parameters (the default-initialised IResNet:
44 of them), `Constant` nodes (roi / scales of opset-11 `Resize`, Reshape shapes), Conv+BN already folded by the exporter, a
`Mul` by a scalar parameter (SCRFD's Scale layer), Dropout dropped at export, BatchNorm1d behind the Gemm, dynamic N / H / W.

CPU: the planner accepts them and the oracle agrees with the torch modules' own float64 outputs (an evaluator that shares nothing with
the oracle).  GPU (`-m gpu`): the HIP path against the oracle and against those float64 outputs.
"""
import os

import numpy as np
import pytest

import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd import api
from oracle import onnx_min, oracle
from tests import util

G = util.GOLDEN
IO = np.load(os.path.join(G, "exported_io.npz"))
HEADS = [f"score_{s}" for s in (8, 16, 32)] + [f"bbox_{s}" for s in (8, 16, 32)] + [f"kps_{s}" for s in (8, 16, 32)]


def _path(name):
    return os.path.join(G, name + ".onnx")


def test_exported_graphs_carry_the_foreign_constructs():
    """The fixtures really contain what the planner used to reject (so the test cannot pass vacuously after a regeneration)."""
    g = onnx_min.load(_path("exported_iresnet_default"))
    ident = [n for n in g.nodes if n.op == "Identity"]
    assert len(ident) >= 10 and all(n.inputs[0] in g.inits for n in ident)          # aliases of initializers
    used = {i for n in g.nodes if n.op in ("Conv", "PRelu", "BatchNormalization") for i in n.inputs[1:]}
    assert used & {n.outputs[0] for n in ident}                                     # ... consumed as weights / slopes / BN vectors
    d = onnx_min.load(_path("exported_scrfd"))
    ops = {n.op for n in d.nodes}
    assert {"Constant", "Resize", "Mul", "Transpose", "Reshape", "Sigmoid"} <= ops
    assert any(n.op == "Conv" and n.attrs.get("group", 1) > 1 for n in d.nodes)


@pytest.mark.parametrize("name,h,w", [("exported_iresnet_default", 112, 112), ("exported_iresnet_trained", 112, 112),
                                      ("exported_scrfd", 96, 128), ("exported_scrfd", 160, 96), ("exported_scrfd", 640, 640)])
def test_planner_accepts_exported_graphs(name, h, w):
    s = api.plan_describe(_path(name), h, w)
    head = s.splitlines()[0]
    assert head.startswith(f"input {h}x{w}")
    outs = [l for l in s.splitlines() if l.startswith("out ")]
    if "scrfd" in name:
        assert [l.split()[1] for l in outs] == HEADS
        a8 = (h // 8) * (w // 8) * 2
        assert outs[0].endswith(f"[{a8}x1]") and outs[3].endswith(f"[{a8}x4]") and outs[6].endswith(f"[{a8}x10]")
        assert " DW+PW " in s or h * w < 640 * 640                                  # the depthwise->pointwise fusion still fires at full size
    else:
        assert outs == ["out embedding [1x32]"]
        assert "+prelu" in s and "+res" in s                                        # fusions survive the alias resolution


@pytest.mark.parametrize("tag", ["default", "trained"])
def test_oracle_matches_torch_fp64_on_exported_iresnet(tag):
    g = onnx_min.load(_path(f"exported_iresnet_{tag}"))
    x, y = IO[f"iresnet_{tag}_x"], IO[f"iresnet_{tag}_y"]
    r = oracle.run_graph(g, {g.inputs[0][0]: x})[g.outputs[0][0]]
    assert r.shape == y.shape
    assert np.abs(r - y).max() < 2e-5 * np.abs(y).max(), np.abs(r - y).max()


@pytest.mark.parametrize("key", ["scrfd", "scrfd2"])
def test_oracle_matches_torch_fp64_on_exported_scrfd(key):
    g = onnx_min.load(_path("exported_scrfd"))
    x = IO[f"{key}_x"]
    out = oracle.run_graph(g, {g.inputs[0][0]: x})
    for nm in HEADS:
        y = IO[f"{key}_{nm}"]
        r = np.asarray(out[nm]).reshape(y.shape)
        assert np.abs(r - y).max() < 2e-5 * max(1.0, np.abs(y).max()), (nm, np.abs(r - y).max())


# ------------------------------------------------------------------------------------------------------------- GPU
def _u8_from_norm(x):
    """The fixtures store (u8 - 127.5) / 128 in RGB planar order; recover the BGR u8 image the path takes."""
    u = np.rint(x * 128.0 + 127.5).astype(np.uint8)
    assert np.array_equal(((u.astype(np.float32) - 127.5) / 128.0), x)
    return np.ascontiguousarray(u.transpose(0, 2, 3, 1)[..., ::-1])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["default", "trained"])
def test_gpu_exported_iresnet_matches_oracle_and_torch_fp64(tag):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "GPU tests need a real device: the product path has no CPU fallback"
    fa.lib().fh_init(0)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(_path(f"exported_iresnet_{tag}")) and orec.loadModel(_path(f"exported_iresnet_{tag}"))
    assert rec.feature_dim() == 32
    x, y = IO[f"iresnet_{tag}_x"], IO[f"iresnet_{tag}_y"]
    crops = _u8_from_norm(x)
    n = len(crops)
    cd = torch.from_numpy(crops).cuda()
    out = torch.zeros((n, 32), device="cuda"); raw = torch.zeros((n, 32), device="cuda")
    assert rec.embed_aligned_dev(cd.data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
    torch.cuda.synchronize()
    got, graw = out.cpu().numpy(), raw.cpu().numpy()
    for i in range(n):
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: oracle.rec_preprocess(crops[i])[None]})[orec.g.outputs[0][0]].reshape(-1)
        assert np.abs(graw[i] - r).max() < 1e-4 * np.abs(r).max()
        assert np.abs(graw[i] - y[i]).max() < 1e-4 * np.abs(y[i]).max()                # the torch module's own float64 output
        ref = oracle.l2_normalize(r)
        assert 1.0 - float(np.dot(got[i], ref)) < 1e-5                                 # north-star bar is 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name,key", [("exported_scrfd_static", "scrfd"), ("exported_scrfd", None)])
def test_gpu_exported_scrfd_heads_match_oracle_and_torch_fp64(name, key):
    """Static export: input size adopted from the file (96 x 128, face_detector.cpp:39-57), heads against the oracle AND the torch
    module's float64 outputs.  Dynamic export: H / W fall back to the reference's 640 x 640 default, heads against the oracle (which
    the CPU tests above pin to torch float64 on the same graph at two other sizes)."""
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available(), "GPU tests need a real device: the product path has no CPU fallback"
    fa.lib().fh_init(0)
    from tests.test_gpu_parity import _det_outputs
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(_path(name)) and odet.loadModel(_path(name))
    if key:
        frame = _u8_from_norm(IO[f"{key}_x"])[0]
        assert det.input_size() == (128, 96)                                     # (width, height)
    else:
        frame = util.frames_u8(1, 640, 640, seed=91, smooth=True)[0]
        assert det.input_size() == (640, 640)
    rows, cols = frame.shape[:2]
    d = torch.from_numpy(frame[None].copy()).cuda()
    assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 1, rows, cols, cols * 3, rows * cols * 3, 0) == 1
    torch.cuda.synchronize()
    got = _det_outputs(det, 1)
    inp, scale = oracle.det_preprocess(frame, odet.inW, odet.inH)
    assert scale == 1.0
    ref = odet.run_network(inp)
    for i, nm in enumerate(HEADS):
        np.testing.assert_allclose(got[i][0], np.asarray(ref[i]).reshape(got[i][0].shape), rtol=1e-4, atol=1e-4, err_msg=nm)
        if key:
            y = IO[f"{key}_{nm}"]
            np.testing.assert_allclose(got[i][0], y.reshape(got[i][0].shape), rtol=1e-4, atol=1e-4, err_msg=nm + " vs torch fp64")
    faces = det.detect_records(frame, 0.5, 0.4)
    oref = odet.detect(frame, 0.5, 0.4)
    assert len(oref) > 0 and abs(len(faces) - len(oref)) <= 1
