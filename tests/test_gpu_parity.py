"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Bars: integer / byte / index work bit-exact; fp32 network outputs within the
tolerances written next to each assert; embeddings within 1e-3 cosine of the oracle.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import facerecognizeonnx_amd as fa            # noqa: E402
from facerecognizeonnx_amd import _lib        # noqa: E402
from oracle import oracle                     # noqa: E402
from tests import util                        # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real device: the product path has no CPU fallback")
    fa.lib().fh_init(0)
    oracle.set_threads(8)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pack_weights(w):
    """ONNX [Cout,Cin,k,k] -> the kernel's packed weight image (layout owned by the library)."""
    cout, cin, k, _ = w.shape
    L = fa.lib()
    rows, kpad = L.fh_conv_wt_rows(cout), L.fh_conv_kpad(k * k * cin)
    out = np.zeros((rows, kpad), np.float32)
    ohwi = np.ascontiguousarray(w.transpose(0, 2, 3, 1))
    assert L.fh_conv_pack_weights(ohwi.ctypes.data, cout, cin, k, out.ctypes.data) == 0
    return out, kpad


CONV_CASES = [
    # B, H,  W,  Cin, Cout, k, stride, cfg
    (2, 14, 14, 64, 128, 3, 1, 0),
    (1, 9, 11, 32, 96, 3, 2, 0),
    (3, 7, 7, 64, 64, 3, 1, 3),
    (1, 28, 28, 64, 64, 3, 2, 1),
    (2, 12, 10, 16, 16, 3, 1, 2),
    (1, 20, 20, 40, 72, 1, 1, 2),
    (1, 13, 13, 4, 24, 3, 2, 2),
    (2, 8, 8, 152, 288, 1, 1, -1),
    (1, 10, 10, 72, 40, 1, 2, -1),
    # >= 512 tiles of 256 rows: the whole tile rounds run in conv_tall_kernel (one LDS image per 32-channel chunk for all nine taps, border
    # taps masked per lane: ragged map, image boundaries inside tiles), the remainder in conv_igemm_kernel with tile0 > 0 (+ its fix-up)
    (66, 40, 52, 32, 48, 3, 1, 1),
    (35, 61, 63, 64, 64, 3, 1, 1),
    # the 128x32 instantiation (SCRFD's merged head convolutions take it at B = 128): 811 tiles, 768 of them in conv_tall_kernel
    (27, 61, 63, 32, 32, 3, 1, 2),
    # 1x1 stride-1 layers with whole 128-row tiles and at least one tile per resident slot: conv_pw_kernel (the lean GEMM form) —
    # 128x96 tiles with K = 288 and with a K tail (152 = 4.75 chunks), 128x32 tiles with N = 152 (padded to 160) and K = 72, 128x64 with K = 40
    (64, 20, 20, 288, 288, 1, 1, -1),
    (64, 20, 20, 152, 288, 1, 1, -1),
    (72, 20, 20, 72, 152, 1, 1, -1),
    (256, 20, 20, 40, 64, 1, 1, -1),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,cfg", CONV_CASES)
def test_conv_layer_matches_oracle(B, H, W, Cin, Cout, k, stride, cfg):
    rng = np.random.default_rng(B * 1000 + H * 10 + Cin)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = oracle.conv2d(x, w, b, stride, k // 2, 1)                       # NCHW
    wp, kpad = pack_weights(w)
    xd, wd, bd = dev(x.transpose(0, 2, 3, 1)), dev(wp), dev(b)
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.full((B, Ho, Wo, Cout), float("nan"), device="cuda")
    rc = fa.lib().fh_conv_forward_dev(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout,
                                      k, stride, kpad, cfg, 0)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy().transpose(0, 3, 1, 2)
    # fp32 sums of <= 2.6k products of O(1) values: 2e-5 absolute covers the order-of-summation difference
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)


DWPW_CASES = [
    # H,  W,  C, Cout, depthwise stride   (maps large enough for the planner to fuse the pair; ragged tiles, channel
    #                                      counts off the 8 / 32 grid)
    (40, 40, 16, 16, 1),                 # stem of 16 channels in front: front_kernel with a stride-1 stem
    (45, 70, 16, 32, 1),                 # ... ragged tiles, row pitch 210 (not a multiple of 4), 32 output channels
    (83, 61, 16, 24, 1),
    (41, 53, 20, 72, 1),
    (48, 40, 40, 40, 1),
    (44, 60, 72, 96, 1),
    (40, 44, 36, 100, 1),
    (56, 40, 100, 24, 1),
    (50, 70, 64, 64, 1),          # the register-fed stride-1 form (dwpw_reg_kernel): C = 16 / 40 / 64 / 72, ragged tiles in x and y
    (40, 40, 16, 64, 1),
    (45, 41, 72, 72, 1),
    (37, 90, 40, 24, 1),
    (160, 160, 16, 40, 2),             # 16 channels, stride 2: the register-fed form with a 17 x 33 halo (dwpw_reg_kernel<.., 2>)
    (163, 175, 16, 40, 2),             # ... odd map: ragged last tile row / column, right / bottom padding inside the halo
    (170, 162, 16, 24, 2),
    (163, 175, 40, 72, 2),
    (161, 166, 20, 100, 2),
]


@pytest.mark.parametrize("H,W,Cc,Cout,ds", DWPW_CASES)
def test_depthwise_pointwise_block_matches_oracle(tmp_path, H, W, Cc, Cout, ds):
    """The fused depthwise 3x3 -> pointwise 1x1 kernel (dwpw_mfma.hip) on a three-conv graph, vs the oracle."""
    from facerecognizeonnx_amd.synth.onnx_writer import OnnxBuilder
    rng = np.random.default_rng(H * 100 + Cc)
    b = OnnxBuilder("dwpw")
    x = b.add_input("input", [1, 3, H, W])
    def conv(x, w, bias, relu=True, **kw):
        y = b.node("Conv", [x, b.init(b.uid("w"), w.astype(np.float32)), b.init(b.uid("b"), bias.astype(np.float32))], **kw)
        return b.node("Relu", [y]) if relu else y
    y = conv(x, rng.standard_normal((Cc, 3, 3, 3)) / 5, rng.standard_normal(Cc) / 10, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1])
    y = conv(y, rng.standard_normal((Cc, 1, 3, 3)) / 3, rng.standard_normal(Cc) / 10, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[ds, ds], group=Cc)
    y = conv(y, rng.standard_normal((Cout, Cc, 1, 1)) / np.sqrt(Cc), rng.standard_normal(Cout) / 10, kernel_shape=[1, 1], strides=[1, 1])
    y = b.node("Transpose", [y], perm=[0, 2, 3, 1])
    b.node("Reshape", [y, b.init("shape", np.array([-1, Cout], np.int64))], outputs=["out"])
    b.add_output("out", ["A", Cout])
    path = b.save(str(tmp_path / "dwpw.onnx"))
    assert "DW+PW" in fa.plan_describe(path, H, W) and ("(depthwise s2)" in fa.plan_describe(path, H, W)) == (ds == 2)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    n = 2
    frames = util.frames_u8(n, H, W, seed=Cout)
    d = dev(frames)
    assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), n, H, W, W * 3, H * W * 3, 0) == n
    torch.cuda.synchronize()
    got = _det_outputs(det, n)[0]
    for i in range(n):
        inp, _ = oracle.det_preprocess(frames[i], W, H)
        ref = odet.run_network(inp)[0]
        np.testing.assert_allclose(got[i], ref.reshape(got[i].shape), rtol=1e-5, atol=2e-5)


WINO_CASES = [
    # B, H,  W,  Cin, Cout      (ragged H / W: tiles of 4x4 outputs hang over the border)
    (2, 14, 14, 256, 256),
    (3, 7, 7, 512, 128),
    (1, 28, 28, 128, 128),
    (2, 13, 10, 128, 64),
    (1, 5, 9, 160, 36),
    (1, 4, 4, 128, 32),
    # B >= 64 on 16-tile maps of side 4 k + 2 / 4 k + 1: the mixed F(4x4) / F(2x2) tiling (four tile classes, planes padded to 128 rows;
    # 16 x 14: only the column direction is mixed, two of the four classes are empty)
    (64, 14, 14, 128, 64),
    (70, 13, 14, 64, 128),
    (64, 16, 14, 64, 64),
]


@pytest.mark.parametrize("B,H,W,Cin,Cout", WINO_CASES)
def test_winograd_conv_matches_oracle(B, H, W, Cin, Cout):
    """Winograd F(4x4,3x3) form of the deep 3x3 stride-1 convolutions (winograd.hip) vs the oracle's direct convolution.
    fp32 Winograd with points 0, +-1, +-2, inf rounds ~25x coarser than the direct form: 2e-4 absolute on O(1) outputs."""
    rng = np.random.default_rng(B * 1000 + H * 10 + Cin)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = oracle.conv2d(x, w, b, 1, 1, 1)
    ohwi = np.ascontiguousarray(w.transpose(0, 2, 3, 1))                    # [Cout][3][3][Cin]
    xd, bd = dev(x.transpose(0, 2, 3, 1)), dev(b)
    out = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    rc = fa.lib().fh_conv_winograd_dev(xd.data_ptr(), ohwi.ctypes.data, bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, 0)
    assert rc == 0, _lib.last_error()
    got = out.cpu().numpy().transpose(0, 3, 1, 2)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4)
    assert np.sqrt(((got - ref) ** 2).mean()) < 2e-5                        # rms error stays at the 1e-5 level


def test_winograd_switch_changes_only_rounding():
    """Full-size IResNet-50 with and without the Winograd form of its 128+-channel 3x3 convolutions."""
    from facerecognizeonnx_amd.synth import models
    rec = fa.FaceRecognizer()
    assert rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
    n = 24
    crops = dev(util.frames_u8(n, 112, 112, seed=21))
    res = []
    for on in (1, 0):
        assert fa.lib().fh_rec_set_winograd(rec.handle, on) == 0
        raw = torch.zeros((n, 512), device="cuda"); out = torch.zeros((n, 512), device="cuda")
        assert rec.embed_aligned_dev(crops.data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
        torch.cuda.synchronize()
        res.append((raw.cpu().numpy().astype(np.float64), out.cpu().numpy().astype(np.float64)))
    (r1, e1), (r0, e0) = res
    assert np.abs(r1 - r0).max() / np.abs(r0).max() < 1e-4                  # raw outputs: relative to their scale
    assert (1.0 - (e1 * e0).sum(1)).max() < 1e-6                            # embeddings: cosine


def test_constant_affine_ops_fold_and_match_oracle(tmp_path):
    """Mul / Add / Sub / Div against constants (scalar and per-channel) are folded into the neighbouring convolutions."""
    from tests.test_host_cpu import _affine_graph
    H, W = 24, 20
    path = _affine_graph(str(tmp_path / "affine.onnx"), H, W)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    frames = util.frames_u8(2, H, W, seed=8)
    d = dev(frames)
    assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 2, H, W, W * 3, H * W * 3, 0) == 2
    torch.cuda.synchronize()
    got = _det_outputs(det, 2)[0]
    for i in range(2):
        inp, _ = oracle.det_preprocess(frames[i], W, H)
        ref = odet.run_network(inp)[0]
        np.testing.assert_allclose(got[i], ref.reshape(got[i].shape), rtol=1e-5, atol=2e-5)


def test_det_preprocess_bit_exact(models_dir):
    det = fa.FaceDetector()
    assert det.loadModel(util.tiny_scrfd(models_dir, hw=128))
    assert fa.lib().fh_det_set_fused_stem(det.handle, 0) == 0       # materialise the preprocessed input tensor
    for rows, cols in ((128, 128), (96, 128), (200, 150), (64, 50)):
        img = util.frames_u8(1, rows, cols, seed=rows, smooth=True)[0]
        ref, scale = oracle.det_preprocess(img, 128, 128)
        d = dev(img)
        rc = fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 1, rows, cols, cols * 3, rows * cols * 3, 0)
        assert rc == 1, _lib.last_error()
        torch.cuda.synchronize()
        p = fa.lib().fh_det_input_dev(det.handle)
        got = _read_dev(p, (128, 128, 4))
        assert np.array_equal(got[..., :3].transpose(2, 0, 1), ref), (rows, cols)
        assert np.all(got[..., 3] == 0)


def _read_dev(ptr, shape):
    out = np.empty(shape, np.float32)
    assert fa.lib().fh_memcpy_d2h(out.ctypes.data, ptr, out.nbytes) == 0, _lib.last_error()
    return out


def _det_outputs(det, n):
    outs = []
    for i in range(fa.lib().fh_det_num_outputs(det.handle)):
        r, c = C.c_int(), C.c_int()
        p = fa.lib().fh_det_output_dev(det.handle, i, C.byref(r), C.byref(c))
        outs.append(_read_dev(p, (n, r.value, c.value)))
    return outs


def test_scrfd_network_and_postprocess(models_dir):
    path = util.tiny_scrfd(models_dir, hw=None, cls_bias=-2.0)            # dynamic H/W -> 640 default
    det = fa.FaceDetector()
    assert det.loadModel(path)
    assert det.input_size() == (640, 640) and det.num_anchors() == 16800
    odet = oracle.OracleDetector()
    assert odet.loadModel(path)
    n = 2
    frames = util.frames_u8(n, 640, 640, seed=5, smooth=True)
    d = dev(frames)
    assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), n, 640, 640, 640 * 3, 640 * 640 * 3, 0) == n
    torch.cuda.synchronize()
    got = _det_outputs(det, n)
    for b in range(n):
        inp, scale = oracle.det_preprocess(frames[b], 640, 640)
        ref = odet.run_network(inp)
        for i in range(9):
            # fp32 network, 40 layers, activations O(1..10): 1e-4 absolute
            np.testing.assert_allclose(got[i][b], ref[i], rtol=1e-4, atol=1e-4, err_msg=f"output {i}")
    # decode + threshold + NMS must be BIT-exact given the same network outputs
    max_pf = 512
    faces = torch.zeros((n, max_pf, 15), dtype=torch.float32, device="cuda")          # 60-byte records
    counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    for thr, nms in ((0.5, 0.4), (0.3, 0.2), (0.45, 0.7), (0.9, 0.4)):
        assert fa.lib().fh_det_postprocess_dev(det.handle, n, thr, nms, faces.data_ptr(), max_pf, counts.data_ptr(), 0) == n
        torch.cuda.synchronize()
        cnt = counts.cpu().numpy()
        rec = faces.cpu().numpy().view(np.uint8).reshape(n, max_pf, 60).copy().view(fa.FACE_DTYPE).reshape(n, max_pf)
        for b in range(n):
            rows = oracle.scrfd_decode([g[b] for g in got], 640, 640)
            ref = oracle.postprocess_rows(rows, 1.0, thr, nms)
            assert cnt[b] == len(ref) and (len(ref) > 0 or thr > 0.5), (thr, nms, cnt[b], len(ref))
            k = min(len(ref), max_pf)
            assert rec[b, :k].tobytes() == ref[:k].tobytes(), (thr, nms, b)


def test_fused_stem_equals_separate_preprocess(models_dir):
    """The u8 -> first-conv fusion must not change results beyond fp32 summation order (letterboxed input included)."""
    det = fa.FaceDetector()
    assert det.loadModel(util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0))
    img = util.frames_u8(1, 90, 128, seed=3, smooth=True)
    d = dev(img)
    outs = []
    for fused in (1, 0):
        assert fa.lib().fh_det_set_fused_stem(det.handle, fused) == 0
        assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 1, 90, 128, 384, 90 * 384, 0) == 1
        outs.append(_det_outputs(det, 1))
    for a, b in zip(*outs):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)


def test_detect_host_api_matches_oracle_end_to_end(models_dir):
    path = util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    for rows, cols in ((128, 128), (100, 180), (300, 200)):
        img = util.frames_u8(1, rows, cols, seed=cols, smooth=True)[0]
        got = det.detect_records(img, 0.5, 0.4)
        ref = odet.detect(img, 0.5, 0.4)
        assert len(ref) > 0
        # network outputs differ by ~1e-6, so a score within that distance of the threshold or a
        # coordinate within that distance of an integer may flip: compare with +-1 px / 1e-4 slack
        # or an IoU within rounding of the NMS threshold may flip: EVERY record is paired (+-1 px / 1e-4), unpaired ones must be such cases
        util.assert_records_equivalent(got, ref, 0.5, 0.4)
    assert len(det.detect_records(None)) == 0
    assert len(det.detect_records(np.zeros((0, 0, 3), np.uint8))) == 0


def test_detect_edge_cases(models_dir):
    """Large non-square frame (strong down-scale + letterbox), a padded row pitch, truncation to the caller's buffer, a threshold
    nothing passes, and a 1 x 1 image — the guards and the un-scale / truncation arithmetic of face_detector.cpp:92-137,249-278."""
    path = util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    big = util.frames_u8(1, 1080, 1920, seed=31, smooth=True)[0]                 # scale = 128/1920: boxes are un-scaled by 15x
    got, ref = det.detect_records(big, 0.5, 0.4), odet.detect(big, 0.5, 0.4)
    assert len(ref) > 0
    # every record paired; box tolerance = one network-output ulp x the 15x un-scale, landmarks likewise (fp32 divide by scale)
    util.assert_records_equivalent(got, ref, 0.5, 0.4, box_tol=15, lm_tol=0.15)
    # the same pixels behind a padded row pitch (cv::Mat::step > cols * 3) give the same records
    padded = np.zeros((1080, 1920 * 3 + 64), np.uint8); padded[:, :1920 * 3] = big.reshape(1080, -1)
    view = np.lib.stride_tricks.as_strided(padded, shape=(1080, 1920, 3), strides=(padded.strides[0], 3, 1))
    assert det.detect_records(view, 0.5, 0.4).tobytes() == got.tobytes()
    # truncation: the best max_faces records, in the same (score-descending) order
    if len(got) > 3:
        assert det.detect_records(big, 0.5, 0.4, max_faces=3).tobytes() == got[:3].tobytes()
    assert len(det.detect_records(big, 1.0, 0.4)) == 0 and len(odet.detect(big, 1.0, 0.4)) == 0       # sigmoid scores never exceed 1
    tiny = np.full((1, 1, 3), 200, np.uint8)
    assert len(det.detect_records(tiny, 0.5, 0.4)) == len(odet.detect(tiny, 0.5, 0.4))


def test_c1_single_jpeg_detect_matches_oracle():
    """BASELINE.json configs[0]: one 640x640 JPEG -> imread -> det_500m detect (full-size synthetic graph)."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    img = fa.imread(os.path.join(os.path.dirname(__file__), "golden", "images", "c1_640x640.jpg"))
    assert img is not None and img.shape == (640, 640, 3)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    got = det.detect_records(img, 0.5, 0.4)
    ref = odet.detect(img, 0.5, 0.4)
    assert len(ref) > 0
    util.assert_records_equivalent(got, ref, 0.5, 0.4)      # every record; only threshold-borderline ones may be unpaired


def test_predecoded_layout_bit_exact(models_dir):
    from facerecognizeonnx_amd.synth import models
    for three_d in (True, False):
        path = models.make_predecoded_det(os.path.join(models_dir, f"pre{int(three_d)}.onnx"), 64, 7, three_d)
        det = fa.FaceDetector(); odet = oracle.OracleDetector()
        assert det.loadModel(path) and odet.loadModel(path)
        img = util.frames_u8(1, 64, 64, seed=9)[0]
        d = dev(img[None])
        assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 1, 64, 64, 192, 64 * 192, 0) == 1
        torch.cuda.synchronize()
        rows = _det_outputs(det, 1)[0][0]
        faces = torch.zeros((1, 64, 15), device="cuda"); counts = torch.zeros(1, dtype=torch.int32, device="cuda")
        assert fa.lib().fh_det_postprocess_dev(det.handle, 1, 0.5, 0.4, faces.data_ptr(), 64, counts.data_ptr(), 0) == 1
        torch.cuda.synchronize()
        ref = oracle.postprocess_rows(rows, 1.0, 0.5, 0.4)
        assert int(counts[0]) == len(ref) and len(ref) > 0
        rec = faces.cpu().numpy().view(np.uint8).reshape(64, 60).copy().view(fa.FACE_DTYPE).reshape(64)
        assert rec[:len(ref)].tobytes() == ref.tobytes()


def _align_gpu(rec, frames, faces, frame_of=None):
    n = len(faces)
    fd, facd = dev(frames), dev(faces.view(np.uint8).reshape(n, 60))
    crops = torch.zeros((n, 112, 112, 3), dtype=torch.uint8, device="cuda")
    ok = torch.zeros(n, dtype=torch.int32, device="cuda")
    fo = dev(np.asarray(frame_of, np.int32)) if frame_of is not None else None
    rows, cols = frames.shape[1:3]
    rc = fa.lib().fh_rec_align_dev(rec.handle, fd.data_ptr(), rows, cols, cols * 3, rows * cols * 3, facd.data_ptr(),
                                   fo.data_ptr() if fo is not None else 0, n, crops.data_ptr(), ok.data_ptr(), 0)
    assert rc == n, _lib.last_error()
    torch.cuda.synchronize()
    return crops.cpu().numpy(), ok.cpu().numpy()


def test_align_bit_exact(models_dir):
    rec = fa.FaceRecognizer()
    assert rec.loadModel(util.tiny_iresnet(models_dir))
    nfr, n = 3, 48
    frames = util.frames_u8(nfr, 240, 320, seed=11, smooth=True)
    lms = util.random_landmarks(n, 240, 320, seed=3)
    rng = np.random.default_rng(4)
    lms[5] = rng.uniform(0, 240, (5, 2))                # garbage landmarks: RANSAC drops points
    lms[6] = lms[6][0]                                   # all coincident -> no transform -> crop fallback
    lms[7] = lms[7][0]
    lms[8] += 400                                        # far outside the frame: all taps read the border
    faces = np.zeros(n, fa.FACE_DTYPE)
    faces["lm"] = lms.reshape(n, 10)
    faces["x"], faces["y"], faces["w"], faces["h"] = 30, 40, 90, 100
    faces[7]["x"] = 1000                                 # fallback with an empty intersection -> empty result
    frame_of = rng.integers(0, nfr, n)
    crops, ok = _align_gpu(rec, frames, faces, frame_of)
    for i in range(n):
        ref = oracle.align_face(frames[frame_of[i]], faces[i])
        if ref is None:
            assert ok[i] == 0, i
        else:
            assert ok[i] in (1, 2), i
            assert np.array_equal(crops[i], ref), f"face {i}: {np.abs(crops[i].astype(int) - ref).max()}"
    assert ok[6] == 2 and ok[7] == 0 and ok[0] == 1


def test_resize_bit_exact():
    rng = np.random.default_rng(0)
    for (sh, sw, dh, dw) in ((480, 640, 112, 112), (100, 100, 50, 50), (37, 53, 112, 112), (640, 360, 640, 360),
                             (123, 77, 61, 200), (64, 64, 128, 128)):
        img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        ref = oracle.resize_bilinear(img, dw, dh)
        s, d = dev(img), torch.zeros((dh, dw, 3), dtype=torch.uint8, device="cuda")
        assert fa.lib().fh_resize_u8c3_dev(s.data_ptr(), sh, sw, sw * 3, d.data_ptr(), dh, dw, dw * 3, 0) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), ref), (sh, sw, dh, dw)


@pytest.mark.parametrize("fold_bn", [True, False])
def test_tiny_iresnet_embeddings(models_dir, fold_bn):
    path = util.tiny_iresnet(models_dir, fold_bn=fold_bn)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    assert rec.feature_dim() == 512
    n = 5
    crops = util.frames_u8(n, 112, 112, seed=2)
    cd = dev(crops)
    out = torch.zeros((n, 512), device="cuda"); raw = torch.zeros((n, 512), device="cuda")
    assert rec.embed_aligned_dev(cd.data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
    torch.cuda.synchronize()
    got, graw = out.cpu().numpy(), raw.cpu().numpy()
    for i in range(n):
        inp = oracle.rec_preprocess(crops[i])
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: inp[None]})[orec.g.outputs[0][0]].reshape(-1)
        np.testing.assert_allclose(graw[i], r, rtol=1e-4, atol=1e-4)
        ref = oracle.l2_normalize(r)
        assert 1.0 - float(np.dot(got[i], ref)) < 1e-5            # north-star bar is 1e-3
        np.testing.assert_allclose(got[i], ref, atol=1e-5)
        assert abs(np.linalg.norm(got[i]) - 1.0) < 1e-5


@pytest.mark.parametrize("fold_bn", [True, False])
def test_tiny_mobilefacenet_embeddings(models_dir, fold_bn):
    """w600k_mbf's op set: grouped 3x3 (2 channels per group), depthwise 3x3 + PReLU, 1x1 expand / project with
    residual, global depthwise conv (GDC), bias-less Linear exported as MatMul, BatchNorm1d."""
    path = util.tiny_mbf(models_dir, fold_bn=fold_bn)
    desc = fa.plan_describe(path, 112, 112)
    assert "GCONV" in desc and "DWGLOBAL" in desc and "DWCONV" in desc and "+prelu" in desc
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    assert rec.feature_dim() == 128
    n = 5
    crops = util.frames_u8(n, 112, 112, seed=4)
    cd = dev(crops)
    out = torch.zeros((n, 128), device="cuda"); raw = torch.zeros((n, 128), device="cuda")
    assert rec.embed_aligned_dev(cd.data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
    torch.cuda.synchronize()
    got, graw = out.cpu().numpy(), raw.cpu().numpy()
    for i in range(n):
        inp = oracle.rec_preprocess(crops[i])
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: inp[None]})[orec.g.outputs[0][0]].reshape(-1)
        np.testing.assert_allclose(graw[i], r, rtol=1e-4, atol=1e-4)
        assert 1.0 - float(np.dot(got[i], oracle.l2_normalize(r))) < 1e-5


def test_full_size_mobilefacenet_cosine():
    from facerecognizeonnx_amd.synth import models
    path = models.cached("w600k_mbf_seed300.onnx", models.make_w600k_mbf)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    n = 3
    crops = util.frames_u8(n, 112, 112, seed=6)
    out = torch.zeros((n, 512), device="cuda")
    assert rec.embed_aligned_dev(dev(crops).data_ptr(), n, out.data_ptr()) == n
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(n):
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: oracle.rec_preprocess(crops[i])[None]})[orec.g.outputs[0][0]].reshape(-1)
        assert 1.0 - float(np.dot(got[i], oracle.l2_normalize(r))) < 1e-5    # north-star bar is 1e-3


def test_extract_feature_host_api(models_dir):
    path = util.tiny_iresnet(models_dir)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    img = util.frames_u8(1, 200, 260, seed=21, smooth=True)[0]
    face = np.zeros(1, fa.FACE_DTYPE)
    face["lm"] = util.random_landmarks(1, 200, 260, seed=8).reshape(1, 10)
    face["x"], face["y"], face["w"], face["h"] = 20, 30, 100, 120
    f1 = rec.extractFeature(img, face[0]); r1 = orec.extractFeature(img, face[0])
    assert f1.shape == (512,) and 1.0 - float(np.dot(f1, r1)) < 1e-5
    f2 = rec.extractFeatureSimple(img); r2 = orec.extractFeatureSimple(img)
    assert 1.0 - float(np.dot(f2, r2)) < 1e-5
    assert abs(rec.compareFaces(f1, f2) - oracle.compare(r1, r2)) < 1e-5
    assert rec.compareFaces(f1, f1[:100]) == 0.0
    assert rec.extractFeature(None, face[0]).size == 0
    bad = face.copy(); bad["lm"] = 5.0; bad["x"] = 5000                   # no transform, empty crop
    assert rec.extractFeature(img, bad[0]).size == 0


def test_full_size_r50_cosine(models_dir):
    """IResNet-50 (6.31 GMAC/face): embeddings within 1e-3 cosine of the oracle (north-star bar)."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    assert abs(rec.macs_per_face() - 6.309e9) < 5e6
    n = 3
    crops = util.frames_u8(n, 112, 112, seed=2, smooth=True)
    cd = dev(crops); out = torch.zeros((n, 512), device="cuda")
    assert rec.embed_aligned_dev(cd.data_ptr(), n, out.data_ptr()) == n
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(2):
        ref = orec.embed_aligned(crops[i])
        cos = float(np.dot(got[i], ref))
        assert 1.0 - cos < 1e-3, cos
        assert np.abs(got[i] - ref).max() < 1e-4
    # batch independence: same crop in a different batch slot gives the same embedding
    cd2 = dev(crops[::-1].copy()); out2 = torch.zeros((n, 512), device="cuda")
    rec.embed_aligned_dev(cd2.data_ptr(), n, out2.data_ptr()); torch.cuda.synchronize()
    np.testing.assert_allclose(out2.cpu().numpy()[::-1], got, atol=1e-6)


def test_pipeline_and_gallery(models_dir):
    dpath = util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)
    rpath = util.tiny_iresnet(models_dir)
    det = fa.FaceDetector(); rec = fa.FaceRecognizer()
    assert det.loadModel(dpath) and rec.loadModel(rpath)
    odet = oracle.OracleDetector(); orec = oracle.OracleRecognizer()
    assert odet.loadModel(dpath) and orec.loadModel(rpath)
    n, F = 6, 2
    frames = util.frames_u8(n, 128, 128, seed=31, smooth=True)
    fd = dev(frames)
    faces = torch.zeros((n * F, 15), device="cuda"); fo = torch.zeros(n * F, dtype=torch.int32, device="cuda")
    emb = torch.zeros((n * F, 512), device="cuda")
    total = fa.pipeline_run_dev(det, rec, fd.data_ptr(), n, 128, 128, F, faces.data_ptr(), fo.data_ptr(), emb.data_ptr())
    torch.cuda.synchronize()
    assert 0 < total <= n * F
    recs = faces.cpu().numpy().view(np.uint8).reshape(n * F, 60).copy().view(fa.FACE_DTYPE).reshape(n * F)[:total]
    frame_of = fo.cpu().numpy()[:total]; e = emb.cpu().numpy()[:total]
    assert np.all(np.diff(frame_of) >= 0)
    for i in range(total):
        ref = orec.extractFeature(frames[frame_of[i]], recs[i])         # oracle on the GPU's own boxes
        assert ref.size == 512 and 1.0 - float(np.dot(e[i], ref)) < 1e-5
    # gallery top-k vs oracle (scores (dot+1)/2, order score desc / index asc)
    rng = np.random.default_rng(4)
    G, k = 5000, 5
    gal = rng.standard_normal((G, 512)).astype(np.float32); gal /= np.linalg.norm(gal, axis=1, keepdims=True)
    gal[123] = e[0]; gal[4000] = e[0]                                     # exact duplicates: tie broken by index
    g = fa.Gallery(512); gd = dev(gal); g.upload(gd.data_ptr(), G, True)
    sc = torch.zeros((total, k), device="cuda"); ix = torch.zeros((total, k), dtype=torch.int32, device="cuda")
    g.topk_dev(emb.data_ptr(), total, k, sc.data_ptr(), ix.data_ptr()); torch.cuda.synchronize()
    rs, ri = oracle.gallery_topk(e, gal, k)
    assert np.array_equal(ix.cpu().numpy(), ri)
    np.testing.assert_allclose(sc.cpu().numpy(), rs, atol=2e-6)
    assert list(ri[0][:2]) == [123, 4000]


def test_gallery_enroll_and_match_labels():
    """The webcam loop's reference handling (main.cpp:229-233,253-256) over an enrolled set: rows are appended one
    call at a time, a query is a Match for its best row iff (dot+1)/2 > threshold (strict), else Unknown (-1)."""
    rng = np.random.default_rng(9)
    unit = lambda a: (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)
    feats = unit(rng.standard_normal((74, 512)))
    q = unit(rng.standard_normal((6, 512)))
    q[0] = feats[5]                                                        # same face: score 1.0
    q[1] = unit(feats[40] + 0.9 * unit(rng.standard_normal(512)))          # cos ~ 0.74 -> mapped ~ 0.87: Match
    q[2] = feats[73]                                                       # a row of the last enrol call
    qd = dev(q)
    lab = torch.full((6,), 7, dtype=torch.int32, device="cuda"); sc = torch.zeros(6, device="cuda")
    g = fa.Gallery(512)
    assert len(g) == 0
    g.label_dev(qd.data_ptr(), 6, 0.6, lab.data_ptr(), sc.data_ptr()); torch.cuda.synchronize()
    assert np.all(lab.cpu().numpy() == -1) and np.all(sc.cpu().numpy() == -1.0)          # nothing enrolled: all Unknown
    assert g.enroll(feats[:1]) == 0 and g.enroll(feats[1:71]) == 1 and g.enroll(feats[71:]) == 71 and len(g) == 74
    g.label_dev(qd.data_ptr(), 6, 0.6, lab.data_ptr(), sc.data_ptr()); torch.cuda.synchronize()
    rs, ri = oracle.gallery_topk(q, feats, 1)
    got_s, got_l = sc.cpu().numpy(), lab.cpu().numpy()
    np.testing.assert_allclose(got_s, rs[:, 0], atol=2e-6)
    assert list(got_l[:3]) == [5, 40, 73]
    assert np.array_equal(got_l, np.where(got_s > np.float32(0.6), ri[:, 0], -1)) and np.any(got_l == -1)
    # strictness: a threshold equal to the score itself is NOT a match, the next float below is
    thr = float(got_s[1])
    g.label_dev(qd.data_ptr(), 6, thr, lab.data_ptr(), sc.data_ptr()); torch.cuda.synchronize()
    assert lab.cpu().numpy()[1] == -1
    g.label_dev(qd.data_ptr(), 6, float(np.nextafter(np.float32(thr), np.float32(-1))), lab.data_ptr(), sc.data_ptr()); torch.cuda.synchronize()
    assert lab.cpu().numpy()[1] == 40


def test_cpp_shim_matches_python_api(models_dir, tmp_path):
    """The reference-shaped C++ classes (shim/face_detector.h, face_recognizer.h) run main.cpp's compare flow."""
    import subprocess
    exe = os.path.join(os.path.dirname(fa.__file__), "shim_demo")
    assert os.path.exists(exe), "shim_demo not built (make -C facerecognizeonnx_amd/csrc)"
    dpath = util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)
    rpath = util.tiny_iresnet(models_dir)
    imgs = util.frames_u8(2, 120, 160, seed=77, smooth=True)
    pa, pb = str(tmp_path / "a.ppm"), str(tmp_path / "b.bmp")             # read back by the shim's cv::imread
    util.write_ppm(pa, imgs[0]); util.write_bmp(pb, imgs[1])
    out = subprocess.run([exe, dpath, rpath, pa, pb, "0.5"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    det = fa.FaceDetector(); rec = fa.FaceRecognizer()
    assert det.loadModel(dpath) and rec.loadModel(rpath)
    fa_, fb_ = det.detect_records(imgs[0]), det.detect_records(imgs[1])
    assert lines["faces"] == f"{len(fa_)} {len(fb_)}" and len(fa_) > 0
    assert [int(v) for v in lines["box"].split()[:4]] == [int(fa_[0][k]) for k in ("x", "y", "w", "h")]
    f1, f2 = rec.extractFeature(imgs[0], fa_[0]), rec.extractFeature(imgs[1], fb_[0])
    assert lines["dim"] == "512 512" and lines["simple"] == "512"
    assert abs(float(lines["similarity"].split()[0]) - rec.compareFaces(f1, f2)) < 2e-6
    assert abs(float(lines["self"]) - 1.0) < 1e-5
    np.testing.assert_allclose([float(v) for v in lines["f1"].split()], f1[:8], atol=2e-6)
    # reference error behaviour through the shim: bad model path -> loadModel false -> exit code -1
    bad = subprocess.run([exe, "/nonexistent.onnx", rpath, pa, pb], capture_output=True, text=True)
    assert bad.returncode != 0 and "Error loading face detector model" in bad.stderr
    bad = subprocess.run([exe, dpath, rpath, str(tmp_path / "missing.jpg"), pb], capture_output=True, text=True)
    assert bad.returncode != 0 and "Cannot read image" in bad.stderr


@pytest.mark.parametrize("n", [37, 96, 128])
def test_streamk_is_deterministic_and_matches_plain_tiles(n):
    """Stream-K remainder round (owner/helper hand-off inside the launch, or slabs + fix-up kernel for the FC):
    bit-identical from run to run, and equal to the plain one-tile-per-workgroup schedule up to fp32 summation
    order (K is cut at different places).  Ragged batch sizes: no stage has a whole number of tile rounds."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)
    rec = fa.FaceRecognizer()
    assert rec.loadModel(path)
    crops = dev(util.frames_u8(n, 112, 112, seed=12))
    res = {}
    for mode in (1, 0, 1, 1):
        assert fa.lib().fh_rec_set_conv_cfg(rec.handle, -1, mode) == 0
        raw = torch.zeros((n, 512), device="cuda"); out = torch.zeros((n, 512), device="cuda")
        assert rec.embed_aligned_dev(crops.data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
        torch.cuda.synchronize()
        r = raw.cpu().numpy()
        assert np.isfinite(r).all()
        if mode in res:
            assert np.array_equal(r, res[mode])
        res[mode] = r
    a, b = res[1].astype(np.float64), res[0].astype(np.float64)
    cos = (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
    assert (1.0 - cos).max() < 1e-6, (1.0 - cos).max()             # different K cut points: fp32 rounding only


def test_remainder_round_for_any_slot_count():
    """fh_rec_set_cus sizes the remainder round for a CU-masked stream; every slot count must give the same
    embeddings up to fp32 summation order (and never dead-lock: helpers are always dispatched before owners)."""
    from facerecognizeonnx_amd.synth import models
    rec = fa.FaceRecognizer()
    assert rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
    n = 48
    crops = dev(util.frames_u8(n, 112, 112, seed=13))
    outs = []
    for cus in (0, 255, 208, 100, 64, 9, 1):
        assert fa.lib().fh_rec_set_cus(rec.handle, cus) == 0
        out = torch.zeros((n, 512), device="cuda")
        assert rec.embed_aligned_dev(crops.data_ptr(), n, out.data_ptr()) == n
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy().astype(np.float64))
    for o in outs[1:]:
        assert (1.0 - (o * outs[0]).sum(1)).max() < 1e-6


def test_cli_modes(models_dir, tmp_path, capsys):
    """Text-mode counterpart of main.cpp's detect / compare / simple modes."""
    from facerecognizeonnx_amd import cli
    dpath = util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)
    rpath = util.tiny_iresnet(models_dir)
    imgs = util.frames_u8(2, 120, 160, seed=77, smooth=True)
    pa, pb = str(tmp_path / "a.ppm"), str(tmp_path / "b.npy")
    util.write_ppm(pa, imgs[0]); np.save(pb, imgs[1])
    assert cli.main(["detect", pa, "--det", dpath]) == 0
    out = capsys.readouterr().out
    det = fa.FaceDetector(); assert det.loadModel(dpath)
    assert f"Detected {len(det.detect(imgs[0]))} faces" in out and "Face 0: box=" in out
    assert cli.main(["compare", pa, pb, "--det", dpath, "--rec", rpath]) == 0
    out = capsys.readouterr().out
    assert "Feature dimension: 512" in out and "Similarity: " in out
    assert cli.main(["simple", pa, pa, "--rec", rpath]) == 0
    assert "Same person" in capsys.readouterr().out                       # identical images -> similarity 1
    assert cli.main(["detect", pa, "--det", str(tmp_path / "missing.onnx")]) == -1
    assert cli.main(["detect", str(tmp_path / "missing.jpg"), "--det", dpath]) == -1


def test_async_two_stream_pipeline_equals_serial(models_dir):
    """fh_pipeline_submit_dev (detector and recogniser on different HIP streams, several batches in flight, the host waiting
    for the detector's face count only) must give exactly the serial entry point's faces and embeddings."""
    det = fa.FaceDetector(); rec = fa.FaceRecognizer()
    assert det.loadModel(util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)) and rec.loadModel(util.tiny_iresnet(models_dir))
    n, F, K = 5, 3, 4
    batches = [dev(util.frames_u8(n, 128, 128, seed=40 + k, smooth=True)) for k in range(K)]
    ref = []
    for k in range(K):
        f = torch.zeros((n * F, 15), device="cuda"); o = torch.zeros(n * F, dtype=torch.int32, device="cuda"); e = torch.zeros((n * F, 512), device="cuda")
        t = fa.pipeline_run_dev(det, rec, batches[k].data_ptr(), n, 128, 128, F, f.data_ptr(), o.data_ptr(), e.data_ptr())
        torch.cuda.synchronize()
        ref.append((t, f[:t].clone(), o[:t].clone(), e[:t].clone()))
    sd, sr = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for k in range(K):
        f = torch.zeros((n * F, 15), device="cuda"); o = torch.zeros(n * F, dtype=torch.int32, device="cuda")
        e = torch.zeros((n * F, 512), device="cuda"); t = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        assert fa.pipeline_submit_dev(det, rec, batches[k].data_ptr(), n, 128, 128, F, f.data_ptr(), o.data_ptr(), e.data_ptr(),
                                      t.data_ptr(), sd.cuda_stream, sr.cuda_stream) == ref[k][0]
        outs.append((t, f, o, e))
    torch.cuda.synchronize()
    for (t, f, o, e), (rt, rf, ro, re_) in zip(outs, ref):
        tt = int(t.item())
        assert tt == rt and rt > 0
        assert torch.equal(f[:tt].view(torch.int32), rf.view(torch.int32))        # 60-byte records, compared bitwise
        assert torch.equal(o[:tt], ro) and torch.equal(e[:tt], re_)


# ---------------------------------------------------------------------------------------------------------------
# Round-2 parity cases: the paths the headline batch really takes, and the branches small inputs never reach.
# ---------------------------------------------------------------------------------------------------------------
def _records(t, n, per):
    return t.cpu().numpy().view(np.uint8).reshape(n, per, 60).copy().view(fa.FACE_DTYPE).reshape(n, per)


@pytest.mark.parametrize("B", [128, 256])
def test_r50_headline_batch_winograd_matches_oracle(B):
    """IResNet-50 at the batch sizes of the headline (128) and of config C2 (256): every 3x3 stride-1 layer with >= 128 input
    channels runs in its Winograd form here (7x7x512 included — it needs B >= 64), which n = 3 never reaches.  Sixteen slots
    spread over the batch (first and last included: every 128-row GEMM tile position, both halves of the stream-K remainder round)
    against the oracle's direct fp32 evaluation (face_recognizer.cpp:279-297)."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    crops = util.frames_u8(B, 112, 112, seed=100 + B)
    out = torch.zeros((B, 512), device="cuda"); raw = torch.zeros((B, 512), device="cuda")
    assert rec.embed_aligned_dev(dev(crops).data_ptr(), B, out.data_ptr(), raw.data_ptr()) == B
    torch.cuda.synchronize()
    got, graw = out.cpu().numpy(), raw.cpu().numpy()
    assert np.isfinite(got).all()
    oracle.set_threads(min(16, os.cpu_count() or 8))
    for i in sorted({int(round(x)) for x in np.linspace(0, B - 1, 16)}):
        inp = oracle.rec_preprocess(crops[i])
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: inp[None]})[orec.g.outputs[0][0]].reshape(-1)
        ref = oracle.l2_normalize(r)
        assert 1.0 - float(np.dot(got[i].astype(np.float64), ref.astype(np.float64))) < 1e-3      # north-star bar
        assert np.abs(got[i] - ref).max() < 1e-4, (i, np.abs(got[i] - ref).max())
        # raw (un-normalised) outputs: fp32 through 50 layers incl. 38 Winograd layers, relative to their scale
        assert np.abs(graw[i] - r).max() < 2e-4 * np.abs(r).max(), (i, np.abs(graw[i] - r).max(), np.abs(r).max())


def test_winograd_bn_link_survives_shortcut_scheduled_first(tmp_path):
    """Winograd conv1 reads the block input's plain tensor and applies bn1 itself; with the shortcut conv written in front of
    conv1 that tensor's last listed reader comes earlier — the arena must not have recycled it."""
    from facerecognizeonnx_amd.synth import models
    path = models.make_iresnet(str(tmp_path / "ds_first.onnx"), (1, 1, 1, 1), (32, 128, 128, 128), 112, 64, seed=5, downsample_first=True)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    n = 24                                                               # 28x28 -> 49 tiles x 24 and 14x14 -> 16 x 24 >= 256: both Winograd
    crops = util.frames_u8(n, 112, 112, seed=77)
    for wino in (1, 0):
        assert fa.lib().fh_rec_set_winograd(rec.handle, wino) == 0
        out = torch.zeros((n, 64), device="cuda"); raw = torch.zeros((n, 64), device="cuda")
        assert rec.embed_aligned_dev(dev(crops).data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
        torch.cuda.synchronize()
        graw = raw.cpu().numpy()
        for i in (0, 11, 23):
            r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: oracle.rec_preprocess(crops[i])[None]})[orec.g.outputs[0][0]].reshape(-1)
            np.testing.assert_allclose(graw[i], r, rtol=2e-4, atol=2e-4 * np.abs(r).max(), err_msg=f"winograd={wino} slot {i}")


def test_rec_preprocess_bit_exact(models_dir):
    """FaceRecognizer::preprocess (face_recognizer.cpp:135-150) stand-alone: BGR->RGB, (v - 127.5f) / 128.0f, planar."""
    rec = fa.FaceRecognizer()
    assert rec.loadModel(util.tiny_iresnet(models_dir))
    assert fa.lib().fh_rec_set_fused_stem(rec.handle, 0) == 0            # materialise the preprocessed input tensor
    n = 3
    crops = util.frames_u8(n, 112, 112, seed=19)
    crops[0, :2, :2] = [[[0, 128, 255], [255, 0, 128]], [[127, 1, 254], [2, 253, 126]]]
    out = torch.zeros((n, 512), device="cuda")
    assert rec.embed_aligned_dev(dev(crops).data_ptr(), n, out.data_ptr()) == n
    torch.cuda.synchronize()
    got = _read_dev(fa.lib().fh_rec_input_dev(rec.handle), (n, 112, 112, 4))
    for i in range(n):
        assert np.array_equal(got[i, ..., :3].transpose(2, 0, 1), oracle.rec_preprocess(crops[i]))
    assert np.all(got[..., 3] == 0)
    assert got[0, 0, 0, 0] == np.float32(0.99609375) and got[0, 0, 0, 2] == np.float32(-0.99609375) and got[0, 0, 0, 1] == np.float32(0.00390625)


def test_det500m_batch8_heads_and_records():
    """Full-size det_500m inside a batch of 8 (config C3's graph): all 9 raw heads of two slots against the oracle's network,
    post-processing bit-exact on the GPU's own heads for every frame, and EVERY post-NMS record of those two slots against
    the oracle's end-to-end detect (+-1 px; a score within 1e-6 of the threshold may enter / leave)."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    n = 8
    frames = np.concatenate([util.frames_u8(4, 640, 640, seed=61), util.frames_u8(4, 640, 640, seed=62, smooth=True)])
    d = dev(frames)
    assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), n, 640, 640, 640 * 3, 640 * 640 * 3, 0) == n
    torch.cuda.synchronize()
    got = _det_outputs(det, n)
    assert [g.shape[1:] for g in got] == [(12800, 1), (3200, 1), (800, 1), (12800, 4), (3200, 4), (800, 4), (12800, 10), (3200, 10), (800, 10)]
    refs = {}
    for b in (0, 7):
        inp, scale = oracle.det_preprocess(frames[b], 640, 640)
        refs[b] = odet.run_network(inp)
        for i in range(9):
            # fp32 through ~50 layers; heads: sigmoid scores in [0,1], distances O(1..10) in stride units
            np.testing.assert_allclose(got[i][b], refs[b][i], rtol=1e-4, atol=1e-4, err_msg=f"slot {b} output {i}")
    max_pf = 1024
    seen = {}
    faces = torch.zeros((n, max_pf, 15), device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    for thr, nms in ((0.5, 0.4), (0.02, 0.4)):                          # the low threshold pushes > 2048 candidates into the NMS
        assert fa.lib().fh_det_postprocess_dev(det.handle, n, thr, nms, faces.data_ptr(), max_pf, counts.data_ptr(), 0) == n
        torch.cuda.synchronize()
        cnt = counts.cpu().numpy(); rec = _records(faces, n, max_pf)
        for b in range(n):
            rows = oracle.scrfd_decode([g[b] for g in got], 640, 640)
            ref = oracle.postprocess_rows(rows, 1.0, thr, nms)
            assert cnt[b] == len(ref), (thr, b, cnt[b], len(ref))
            k = min(len(ref), max_pf)
            assert rec[b, :k].tobytes() == ref[:k].tobytes(), (thr, b)
            seen[thr] = max(seen.get(thr, 0), len(oracle.threshold_rows(rows, 1.0, thr)))
    assert seen[0.5] > 0 and seen[0.02] > 2048, seen                    # the low threshold really took the global-memory NMS branch
    # end to end for the two oracle slots: every record must have its counterpart
    assert fa.lib().fh_det_postprocess_dev(det.handle, n, 0.5, 0.4, faces.data_ptr(), max_pf, counts.data_ptr(), 0) == n
    torch.cuda.synchronize()
    cnt = counts.cpu().numpy(); rec = _records(faces, n, max_pf)
    for b in (0, 7):
        ref = odet.detect(frames[b], 0.5, 0.4)
        g = rec[b, :cnt[b]]
        assert abs(len(g) - len(ref)) <= 2, (len(g), len(ref))
        used = np.zeros(len(g), bool)
        missing = 0
        for r in ref:
            near = np.where(~used & (np.abs(g["score"] - r["score"]) < 1e-4))[0]
            ok = [j for j in near if max(abs(int(g[j][k]) - int(r[k])) for k in ("x", "y", "w", "h")) <= 1 and
                  np.abs(g[j]["lm"] - r["lm"]).max() < 1e-2]
            if ok:
                used[ok[0]] = True
            else:
                missing += 1
        assert missing <= 2 and (~used).sum() <= 2, (b, missing, (~used).sum())


def _crafted_rows(rng, R, n_live, feat=15, thr=0.5):
    """Pre-decoded rows x1,y1,x2,y2,score,kps with the corner cases of face_detector.cpp:249-384 planted in."""
    rows = np.zeros((R, feat), np.float32)
    rows[:, 4] = rng.uniform(0.0, thr, R)                               # dead by default
    live = rng.permutation(R)[:n_live]
    x1 = rng.uniform(-40, 560, n_live); y1 = rng.uniform(-40, 560, n_live)
    w = rng.uniform(8, 120, n_live); h = rng.uniform(8, 120, n_live)
    rows[live, 0], rows[live, 1], rows[live, 2], rows[live, 3] = x1, y1, x1 + w, y1 + h
    rows[live, 4] = np.floor(rng.uniform(thr, 1.0, n_live) * 64 + 1) / 64      # 1/64 grid: exact score ties, all > thr
    rows[live, 5:15] = rng.uniform(-20, 660, (n_live, 10))
    if n_live >= 64:
        z = live[:24]
        rows[z[:8], 2] = rows[z[:8], 0]                                 # zero width  (0/0 IoU between two of them: NaN, never suppressed)
        rows[z[8:16], 3] = rows[z[8:16], 1]                             # zero height
        rows[z[16:20], 0:4] = [100.25, 100.75, 100.25, 100.75]          # identical zero-area boxes
        rows[z[20:24], 0:4] = [-0.5, -0.9, 30.4, 30.6]                  # int(-0.5) = 0: truncation toward zero
        d = live[24:40]
        rows[d[1::2]] = rows[d[0::2]]                                   # exact duplicates: IoU 1, tie broken by row index
    if feat > 15:
        rows[:, 15:] = rng.standard_normal((R, feat - 15))              # extra columns are ignored
    dead = np.setdiff1d(np.arange(R), live)
    if len(dead) >= 8:
        rows[dead[0], 4] = np.nan                                       # NaN score: not > thr
        rows[dead[1], 4] = thr                                          # strict >
        rows[dead[2], 4] = -np.inf
        rows[dead[3], 4] = np.nextafter(np.float32(thr), np.float32(0))
    return rows


@pytest.mark.parametrize("feat,scale", [(15, 1.0), (17, 0.37)])
def test_postprocess_crafted_rows_bit_exact_both_nms_branches(feat, scale):
    """rows_threshold_kernel + sort_nms_kernel through fh_postprocess_rows_dev on crafted rows: frames with 0, a few hundred,
    exactly 2048 (LDS branch), 2049 and 5000 survivors (global-memory sort + sweep), score ties, zero-area boxes (0/0 IoU),
    negative coordinates, NaN / -inf / == threshold scores, max_out truncation.  Bit-exact against the oracle."""
    rng = np.random.default_rng(int(scale * 100) + feat)
    R, thr = 6000, 0.5
    lives = [0, 300, 2048, 2049, 5000, 1]
    rows = np.stack([_crafted_rows(rng, R, nl, feat, thr) for nl in lives])
    n = len(lives)
    d = dev(rows)
    for nms, max_pf in ((0.4, 6000), (0.1, 6000), (0.4, 100)):
        faces = torch.zeros((n, max_pf, 15), device="cuda"); counts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        rc = fa.lib().fh_postprocess_rows_dev(d.data_ptr(), n, R, feat, scale, thr, nms, faces.data_ptr(), max_pf, counts.data_ptr(), 0)
        assert rc == n, _lib.last_error()
        torch.cuda.synchronize()
        cnt = counts.cpu().numpy(); rec = _records(faces, n, max_pf)
        for b in range(n):
            pre = oracle.threshold_rows(rows[b], scale, thr)
            assert len(pre) == lives[b]
            ref = oracle.nms(pre, nms) if len(pre) else pre
            assert cnt[b] == len(ref), (b, nms, cnt[b], len(ref))
            k = min(len(ref), max_pf)
            assert rec[b, :k].tobytes() == ref[:k].tobytes(), (b, nms, max_pf)
            if k:
                assert np.all(np.diff(ref["score"]) <= 0)                # score-descending, as main.cpp:101-104 relies on
    # feat < 15: "Unexpected output shape format" -> no boxes (face_detector.cpp:300-303)
    faces = torch.zeros((n, 8, 15), device="cuda"); counts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    assert fa.lib().fh_postprocess_rows_dev(d.data_ptr(), n, R, 14, scale, thr, 0.4, faces.data_ptr(), 8, counts.data_ptr(), 0) == n
    torch.cuda.synchronize()
    assert np.all(counts.cpu().numpy() == 0)


def test_iou_zero_over_zero_and_strictness_on_gpu():
    """FaceDetector::iou corner cases (face_detector.cpp:340-354,369-371) as they come out of the GPU NMS: two zero-area boxes
    (0/0 = NaN, not suppressed), IoU exactly equal to the threshold (strict >, not suppressed), just above (suppressed)."""
    def rows_of(boxes, scores):
        r = np.zeros((len(boxes), 15), np.float32)
        for i, ((x, y, w, h), s) in enumerate(zip(boxes, scores)):
            r[i, :5] = [x, y, x + w, y + h, s]
        return r
    cases = [
        ([(10, 10, 0, 0), (10, 10, 0, 0)], [0.9, 0.8], 0.4, 2),                 # 0/0
        ([(0, 0, 10, 10), (0, 0, 10, 5)], [0.9, 0.8], 0.5, 2),                  # IoU = 50/100 = 0.5, thr 0.5: kept
        ([(0, 0, 10, 10), (0, 0, 10, 5)], [0.9, 0.8], 0.49, 1),                 # suppressed
        ([(0, 0, 10, 10), (20, 20, 5, 5), (0, 0, 10, 10)], [0.7, 0.8, 0.7], 0.4, 2),   # tie: lower row index first, duplicate suppressed
    ]
    for boxes, scores, nms, want in cases:
        rows = rows_of(boxes, scores)[None]
        faces = torch.zeros((1, 8, 15), device="cuda"); counts = torch.zeros(1, dtype=torch.int32, device="cuda")
        assert fa.lib().fh_postprocess_rows_dev(dev(rows).data_ptr(), 1, len(boxes), 15, 1.0, 0.5, nms, faces.data_ptr(), 8, counts.data_ptr(), 0) == 1
        torch.cuda.synchronize()
        ref = oracle.postprocess_rows(rows[0], 1.0, 0.5, nms)
        assert int(counts[0]) == want == len(ref)
        assert _records(faces, 1, 8)[0, :want].tobytes() == ref.tobytes()


def test_host_image_view_at_the_end_of_its_buffer(models_dir):
    """cv::Mat ROI / numpy column slice: the last row owns cols*3 bytes, not a whole pitch: the host-pointer
    entry points must not read past it, and must give the records of the contiguous copy."""
    import mmap
    det = fa.FaceDetector(); rec = fa.FaceRecognizer()
    assert det.loadModel(util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)) and rec.loadModel(util.tiny_iresnet(models_dir))
    rows, cols, pitch = 90, 100, 160 * 3
    img = util.frames_u8(1, rows, cols, seed=5, smooth=True)[0]
    need = (rows - 1) * pitch + cols * 3
    size = (need + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE
    mm = mmap.mmap(-1, size)
    buf = np.frombuffer(mm, np.uint8)
    view = np.lib.stride_tricks.as_strided(buf[size - need:], shape=(rows, cols, 3), strides=(pitch, 3, 1), writeable=True)
    view[...] = img                                                      # the last pixel is the last byte of the mapping
    a, b = det.detect_records(view, 0.5, 0.4), det.detect_records(img, 0.5, 0.4)
    assert len(b) > 0 and a.tobytes() == b.tobytes()
    assert np.array_equal(rec.extractFeatureSimple(view), rec.extractFeatureSimple(img))
    assert np.array_equal(rec.extractFeature(view, b[0]), rec.extractFeature(img, b[0]))
    del view, buf


@pytest.mark.parametrize("Q,k", [(1, 1), (48, 16), (256, 5)])
def test_gallery_beyond_one_slab(Q, k):
    """1:N compareFaces (face_recognizer.cpp:320-334 generalised) on a gallery of 2^20 + 4097 rows — beyond one 2^20-row slab,
    the size a 10 M gallery sharded over 8 ranks needs — with planted exact duplicates across the slab boundary."""
    rng = np.random.default_rng(Q)
    G = (1 << 20) + 4097
    gal = rng.standard_normal((G, 512), dtype=np.float32)
    gal /= np.linalg.norm(gal, axis=1, keepdims=True)
    q = rng.standard_normal((Q, 512)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    gal[5] = q[0]; gal[(1 << 20) - 1] = q[0]; gal[1 << 20] = q[0]; gal[G - 1] = q[0]      # ties on both sides of the boundary
    g = fa.Gallery(512)
    gd = dev(gal); g.upload(gd.data_ptr(), G, True, 1000)                # global index base of a shard
    del gd
    sc = torch.zeros((Q, k), device="cuda"); ix = torch.full((Q, k), -7, dtype=torch.int32, device="cuda")
    g.topk_dev(dev(q).data_ptr(), Q, k, sc.data_ptr(), ix.data_ptr()); torch.cuda.synchronize()
    rs, ri = oracle.gallery_topk(q, gal, k)
    gi, gs = ix.cpu().numpy() - 1000, sc.cpu().numpy()
    np.testing.assert_allclose(gs, rs, atol=2e-6)
    # indices: identical, except that two DIFFERENT rows whose scores differ by less than the fp32 summation-order noise of a
    # 512-term dot (the oracle sums sequentially, the GPU in MFMA order) may swap ranks — judged on exact fp64 scores
    exact = lambda idx: ((np.einsum("qkd,qd->qk", gal[idx].astype(np.float64), q.astype(np.float64))) + 1.0) / 2.0
    eg, er = exact(gi), exact(ri)
    diff = gi != ri
    assert np.abs(eg - er)[diff].max(initial=0.0) < 1e-6 and diff.mean() < 0.01, (diff.sum(), np.abs(eg - er)[diff].max(initial=0.0))
    assert np.all(np.diff(eg, axis=1) <= 1e-6)
    # exact ties (identical rows) are ordered by global index, across the slab boundary too
    assert list(ix.cpu().numpy()[0][:min(k, 4)]) == [1005, 1000 + (1 << 20) - 1, 1000 + (1 << 20), 1000 + G - 1][:min(k, 4)]


def test_sharded_gallery_merge_kernel_equals_single_gallery():
    """Row-sharded gallery: per-shard top-k lists merged by fh_topk_merge_dev give exactly the single-gallery
    answer, duplicates on different shards included (index tie-break), also when a shard holds fewer than k rows."""
    from facerecognizeonnx_amd.distributed import merge_topk_dev, shard_range
    rng = np.random.default_rng(12)
    G, Q, k = 30011, 37, 16
    gal = rng.standard_normal((G, 512)).astype(np.float32); gal /= np.linalg.norm(gal, axis=1, keepdims=True)
    q = gal[rng.integers(0, G, Q)] + 0.3 * rng.standard_normal((Q, 512)).astype(np.float32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    gal[100] = q[0]; gal[20000] = q[0]; gal[G - 1] = q[0]
    qd = dev(q)
    whole = fa.Gallery(512); whole.upload(dev(gal).data_ptr(), G, True)
    ws = torch.zeros((Q, k), device="cuda"); wi = torch.zeros((Q, k), dtype=torch.int32, device="cuda")
    whole.topk_dev(qd.data_ptr(), Q, k, ws.data_ptr(), wi.data_ptr())
    for world in (2, 8):
        bounds = [shard_range(G, r, world) for r in range(world)]
        bounds[-1] = (G - 5, G); bounds[-2] = (bounds[-2][0], G - 5)      # a last shard with only 5 (< k) rows
        ps = torch.zeros((world, Q, k), device="cuda"); pi = torch.zeros((world, Q, k), dtype=torch.int32, device="cuda")
        for r, (b, e) in enumerate(bounds):
            g = fa.Gallery(512); g.upload(dev(gal[b:e]).data_ptr(), e - b, True, b)
            g.topk_dev(qd.data_ptr(), Q, k, ps[r].data_ptr(), pi[r].data_ptr())
        ms, mi = merge_topk_dev(ps, pi, k)
        torch.cuda.synchronize()
        assert torch.equal(mi, wi) and torch.equal(ms, ws), world
    assert list(wi.cpu().numpy()[0][:3]) == [100, 20000, G - 1]


def test_bench_two_ranks_on_one_gpu_through_the_launcher():
    """`bench.py --gpus 2` started WITHOUT a launcher: it must spawn its two ranks itself (gloo + both on cuda:0 here — the
    box has one GPU), shard the gallery, all-gather queries + per-rank top-k and merge on the GPU, and print ONE line."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--all-ranks-on-device0",
                          "--frames", "8", "--steps", "2", "--warmup", "1", "--gallery", "50000", "--no-cpu-baseline", "--no-kernel-timing"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["gallery_rows"] == 50000
    assert abs(d["value"] - 2 * 8 * 2 / (d["ms_per_step"] * 2 * 1e-3)) / d["value"] < 1e-6      # faces of BOTH ranks / max-over-ranks time


def test_pipeline_embeds_live_faces_only(models_dir):
    """The reference embeds "for every face" (main.cpp:221-238): 0..F per frame.  The pipeline must hand the recogniser exactly
    the live faces (compacted, frame order kept), give the serial per-face API's embeddings, and spend recogniser time in
    proportion to them — a batch of empty frames costs (almost) nothing."""
    from facerecognizeonnx_amd.synth import models
    det = fa.FaceDetector(); rec = fa.FaceRecognizer()
    assert det.loadModel(util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0))
    assert rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
    n, F, NMS = 32, 4, 1.0                                              # IoU never exceeds 1: nothing is suppressed, neighbouring anchors all count
    frames = util.frames_u8(n, 128, 128, seed=90, smooth=True)
    # pick a threshold that leaves a MIX of 0..F faces per frame: the 40th percentile of the frames' best scores
    allf = torch.zeros((n, 64, 15), device="cuda"); cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    fd = dev(frames)
    det.detect_batch_dev(fd.data_ptr(), n, 128, 128, allf.data_ptr(), 64, cnt.data_ptr(), 0.3, NMS)
    torch.cuda.synchronize()
    recs = _records(allf, n, 64); c0 = cnt.cpu().numpy()
    assert c0.min() >= 1
    top = sorted(float(recs[b, 0]["score"]) for b in range(n))
    thr = top[int(0.4 * n)] - 1e-6
    det.detect_batch_dev(fd.data_ptr(), n, 128, 128, allf.data_ptr(), 64, cnt.data_ptr(), thr, NMS)
    torch.cuda.synchronize()
    want = np.minimum(cnt.cpu().numpy(), F)
    assert want.min() == 0 and want.max() >= 2 and 0 < want.sum() < n * F, want
    faces = torch.zeros((n * F, 15), device="cuda"); fo = torch.full((n * F,), -1, dtype=torch.int32, device="cuda")
    emb = torch.zeros((n * F, 512), device="cuda")
    total = fa.pipeline_run_dev(det, rec, fd.data_ptr(), n, 128, 128, F, faces.data_ptr(), fo.data_ptr(), emb.data_ptr(), thr, NMS)
    torch.cuda.synchronize()
    assert total == int(want.sum())
    got_fo = fo.cpu().numpy()[:total]
    assert np.array_equal(got_fo, np.repeat(np.arange(n), want))         # compacted, frame order, min(count, F) each
    # serial API on the same faces: one face at a time through fh_rec_embed_faces_dev
    ser = torch.zeros((total, 512), device="cuda")
    for i in range(total):
        one = faces[i:i + 1].contiguous(); fo1 = fo[i:i + 1].contiguous()
        assert fa.lib().fh_rec_embed_faces_dev(rec.handle, fd.data_ptr(), 128, 128, 384, 128 * 384, one.data_ptr(), fo1.data_ptr(), 1,
                                               ser[i:i + 1].data_ptr(), 0, 0) == 1
    torch.cuda.synchronize()
    a, b = emb[:total].cpu().numpy().astype(np.float64), ser.cpu().numpy().astype(np.float64)
    assert (1.0 - (a * b).sum(1)).max() < 1e-6                           # batch of `total` vs batch of 1: fp32 summation order only
    assert torch.all(emb[total:] == 0)                                   # slots beyond the live faces are never written

    def timed(thr_):
        for _ in range(2):
            t = fa.pipeline_run_dev(det, rec, fd.data_ptr(), n, 128, 128, F, faces.data_ptr(), fo.data_ptr(), emb.data_ptr(), thr_, NMS)
        torch.cuda.synchronize()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        for _ in range(5):
            fa.pipeline_run_dev(det, rec, fd.data_ptr(), n, 128, 128, F, faces.data_ptr(), fo.data_ptr(), emb.data_ptr(), thr_, NMS)
        b_.record(); torch.cuda.synchronize()
        return t, a_.elapsed_time(b_) / 5
    t_full, ms_full = timed(0.05)                                        # (nearly) every frame has >= F faces
    t_none, ms_none = timed(1.0)                                         # nothing passes
    t_mix, ms_mix = timed(thr)
    assert t_none == 0 and t_full > 2 * t_mix > 0
    assert ms_none < 0.25 * ms_full and ms_mix < 0.75 * ms_full, (ms_none, ms_mix, ms_full, t_mix, t_full)


def test_frame_stream_equals_device_pipeline(models_dir):
    """fh_stream_* (host frames, uploads overlapped with the previous batch, results one batch later) returns exactly what
    fh_pipeline_run_dev returns for the same frames resident in HBM; ring-full / empty-ring states are errors, not hangs."""
    det = fa.FaceDetector(); rec = fa.FaceRecognizer()
    assert det.loadModel(util.tiny_scrfd(models_dir, hw=128, cls_bias=-2.0)) and rec.loadModel(util.tiny_iresnet(models_dir))
    n, F, K = 6, 2, 5
    batches = [util.frames_u8(n if k != 3 else 4, 128, 128, seed=70 + k, smooth=True) for k in range(K)]      # one short batch
    ref = []
    for fr in batches:
        m = len(fr)
        f = torch.zeros((m * F, 15), device="cuda"); o = torch.zeros(m * F, dtype=torch.int32, device="cuda"); e = torch.zeros((m * F, 512), device="cuda")
        t = fa.pipeline_run_dev(det, rec, dev(fr).data_ptr(), m, 128, 128, F, f.data_ptr(), o.data_ptr(), e.data_ptr())
        torch.cuda.synchronize()
        ref.append((t, _records(f.view(1, m * F, 15), 1, m * F)[0][:t], o.cpu().numpy()[:t], e.cpu().numpy()[:t]))
    st = fa.FrameStream(det, rec, n, 128, 128, F)
    with pytest.raises(fa.FaceHipError):
        st.collect()                                                     # nothing in flight
    got = []
    assert st.submit(batches[0]) == ref[0][0] and st.submit(batches[1]) == ref[1][0]      # fill the ring ...
    with pytest.raises(fa.FaceHipError):
        st.submit(batches[2])                                            # ... a third batch in flight must be refused
    for k in range(2, K):
        got.append(st.collect())                                         # batch k-2, while batch k-1 is still in flight
        assert st.submit(batches[k]) == ref[k][0]
    got.append(st.collect()); got.append(st.collect())
    assert len(got) == K
    for (t, rf, ro, re_), (gf, go, ge) in zip(ref, got):
        assert len(gf) == t and gf.tobytes() == rf.tobytes() and np.array_equal(go, ro) and np.array_equal(ge, re_)
    st.close()


def test_det500m_letterboxed_frame_heads_match_oracle():
    """det_500m on a frame that is NOT 640x640 (500 x 375 -> scale 1.28, resized to 640 x 480, zero canvas below): the
    thread-per-pixel stem kernel's interior / border split with a letterbox (u8 zeros, not conv padding) — all 9 heads vs oracle."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    # (640 x 3 / x 2 / x 1 and 2 x 640: scale 1, a sliver pasted into the canvas — row-end dwords of rows only 9 / 6 / 3 bytes long; the
    #  one-pixel-wide frame takes the separate stem kernel)
    for rows, cols in ((375, 500), (640, 401), (640, 3), (640, 2), (640, 1), (2, 640)):
        img = util.frames_u8(1, rows, cols, seed=rows + cols, smooth=min(rows, cols) >= 32)
        d = dev(img)
        assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 1, rows, cols, cols * 3, rows * cols * 3, 0) == 1
        torch.cuda.synchronize()
        got = _det_outputs(det, 1)
        inp, scale = oracle.det_preprocess(img[0], 640, 640)
        ref = odet.run_network(inp)
        for i in range(9):
            np.testing.assert_allclose(got[i][0], ref[i], rtol=1e-4, atol=1e-4, err_msg=f"{rows}x{cols} output {i}")


def test_det500m_seeded_random_frame_shapes_and_pitches_match_oracle():
    """The fused front kernel's window staging (row-end dwords read from a clamped position and shifted into place, letterbox canvas,
    conv padding patched in on border tiles) on frame shapes nobody picked by hand: 10 seeded (rows, cols) incl. odd widths, shapes
    that are resized first, and padded row pitches off the dword grid — heads of det_500m vs the oracle."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    rng = np.random.default_rng(2026)
    shapes = [(int(rng.integers(33, 900)), int(rng.integers(33, 900))) for _ in range(8)] + [(640, 639), (639, 640)]
    for k, (rows, cols) in enumerate(shapes):
        img = util.frames_u8(1, rows, cols, seed=rows * 7 + cols, smooth=True)
        pad = int(rng.integers(0, 7)) if k % 2 else 0                    # every other frame: a row pitch with 0..6 bytes of padding
        step = cols * 3 + pad
        buf = np.zeros((rows, step), np.uint8); buf[:, :cols * 3] = img[0].reshape(rows, cols * 3)
        d = dev(buf)
        assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 1, rows, cols, step, rows * step, 0) == 1
        torch.cuda.synchronize()
        got = _det_outputs(det, 1)
        inp, scale = oracle.det_preprocess(img[0], 640, 640)
        ref = odet.run_network(inp)
        for i in range(9):
            np.testing.assert_allclose(got[i][0], ref[i], rtol=1e-4, atol=1e-4, err_msg=f"{rows}x{cols} step {step} output {i}")


def test_fused_front_equals_separate_kernels():
    """Stem conv computed inside the first depthwise -> pointwise kernel (SCRFD's opening block) vs the three-kernel form: same heads
    up to fp32 summation order, on a full frame, a letterboxed frame and a frame with a padded row pitch."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    det = fa.FaceDetector()
    assert det.loadModel(path)
    assert "0 CONV k3s2 640x640x4 -> 320x320x16+relu" in fa.plan_describe(path, 640, 640) and "1 DW+PW k1s1 320x320x16" in fa.plan_describe(path, 640, 640)
    for rows, cols, pitch in ((640, 640, 1920), (333, 640, 1920), (640, 250, 250 * 3 + 5)):
        img = np.zeros((2, rows, pitch), np.uint8)
        img[:, :, :cols * 3] = util.frames_u8(2, rows, cols, seed=rows + cols).reshape(2, rows, cols * 3)
        d = dev(img)
        outs = []
        for on in (1, 0):
            assert fa.lib().fh_det_set_fused_front(det.handle, on) == 0
            assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 2, rows, cols, pitch, rows * pitch, 0) == 2
            torch.cuda.synchronize()
            outs.append(_det_outputs(det, 2))
        for a, b in zip(*outs):
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5, err_msg=f"{rows}x{cols} pitch {pitch}")


def test_fused_winograd_transforms_equal_separate_kernels():
    """Output transform of one Winograd layer + input transform of the next as ONE kernel (the activation stays in LDS) vs the two
    separate kernels: same arithmetic on the same values, so the raw network outputs must agree to fp32 rounding — on the 14x14 and
    7x7 stages of the full IResNet-50 (conv1 -> conv2 inside a block, conv2 -> next block's conv1 through its BatchNorm + residual)
    and with a batch that does not fill the last 4-image group of the 7x7 kernel."""
    from facerecognizeonnx_amd.synth import models
    rec = fa.FaceRecognizer()
    assert rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
    for n in (130, 67):                                               # 7x7 needs B >= 64 for Winograd; 130 / 67 are not multiples of 4
        crops = dev(util.frames_u8(n, 112, 112, seed=300 + n))
        res = []
        for on in (1, 0):
            assert fa.lib().fh_rec_set_wino_fusion(rec.handle, on) == 0
            raw = torch.zeros((n, 512), device="cuda"); out = torch.zeros((n, 512), device="cuda")
            assert rec.embed_aligned_dev(crops.data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
            torch.cuda.synchronize()
            res.append(raw.cpu().numpy().astype(np.float64))
        assert np.isfinite(res[0]).all()
        assert np.abs(res[0] - res[1]).max() <= 1e-5 * np.abs(res[1]).max(), np.abs(res[0] - res[1]).max()


HALO_CASES = [
    # H,  W,  Cin, Cout, residual   (ragged tiles: H % 8, W % 16 != 0; Cout off the 4 / 32 grid; both n-tile counts)
    (80, 80, 16, 16, False),
    (40, 40, 16, 16, True),
    (21, 37, 16, 16, True),
    (80, 80, 16, 30, False),
    (23, 50, 16, 30, False),
    (40, 24, 16, 64, True),
    (33, 17, 16, 40, False),
    (19, 33, 16, 8, True),            # Cout <= 16: the 16x16x4-MFMA form (conv3x3_halo16_kernel), output channels 8 / 12 of 16 rows live
    (40, 40, 16, 12, False),
]


@pytest.mark.parametrize("H,W,Cin,Cout,res", HALO_CASES)
def test_halo_conv_matches_oracle(tmp_path, H, W, Cin, Cout, res):
    """Spatial-tile 3x3 convolution (conv_halo.hip) inside a small graph — stem conv -> [3x3 (-> ReLU) (+ residual)] — vs the oracle,
    and vs the generic implicit-GEMM kernel on the same graph."""
    from facerecognizeonnx_amd.synth.onnx_writer import OnnxBuilder
    rng = np.random.default_rng(H * 100 + Cin + Cout)
    b = OnnxBuilder("halo")
    x = b.add_input("input", [1, 3, H, W])
    def conv(x, cout, cin, k, relu):
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        y = b.node("Conv", [x, b.init(b.uid("w"), w), b.init(b.uid("b"), (rng.standard_normal(cout) / 10).astype(np.float32))],
                   kernel_shape=[k, k], pads=[k // 2] * 4, strides=[1, 1])
        return b.node("Relu", [y]) if relu else y
    y0 = conv(x, Cin, 3, 3, True)
    if res:                                                  # the residual must have Cout channels: a 1x1 projection of the stem map
        side = conv(y0, Cout, Cin, 1, False)
        y = b.node("Add", [conv(y0, Cout, Cin, 3, False), side])
    else:
        y = conv(y0, Cout, Cin, 3, True)
    y = b.node("Transpose", [y], perm=[0, 2, 3, 1])
    b.node("Reshape", [y, b.init("shape", np.array([-1, Cout], np.int64))], outputs=["out"])
    b.add_output("out", ["A", Cout])
    path = b.save(str(tmp_path / "halo.onnx"))
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    n = 3
    frames = util.frames_u8(n, H, W, seed=Cout)
    d = dev(frames)
    outs = []
    for on in (1, 0):
        assert fa.lib().fh_det_set_halo_conv(det.handle, on) == 0
        assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), n, H, W, W * 3, H * W * 3, 0) == n
        torch.cuda.synchronize()
        outs.append(_det_outputs(det, n)[0])
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-5, atol=2e-5)
    for i in range(n):
        inp, _ = oracle.det_preprocess(frames[i], W, H)
        ref = odet.run_network(inp)[0]
        np.testing.assert_allclose(outs[0][i], ref.reshape(outs[0][i].shape), rtol=1e-5, atol=3e-5)


@pytest.mark.timeout(300)
def test_two_streams_full_size_models_remainder_rounds():
    """Full-size det_500m and IResNet-50 on two HIP streams at once (fh_pipeline_submit_dev, three batches in flight), with batch sizes
    whose convolutions all end in a stream-K remainder round (owner workgroups wait for helper workgroups of the SAME launch): the
    helpers have lower block indices and are therefore dispatched first, so an owner can never wait for a helper that has no slot —
    also when another stream's kernels occupy part of the chip.  Results must equal the serial entry point bit for bit."""
    from facerecognizeonnx_amd.synth import models
    det = fa.FaceDetector(); rec = fa.FaceRecognizer()
    assert det.loadModel(models.cached("det_500m_seed100.onnx", models.make_det_500m))
    assert rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
    n, F, K = 23, 3, 3                                                   # 69 faces: ragged tile counts in every stage
    batches = [dev(util.frames_u8(n, 640, 640, seed=500 + k)) for k in range(K)]
    ref = []
    for k in range(K):
        f = torch.zeros((n * F, 15), device="cuda"); o = torch.zeros(n * F, dtype=torch.int32, device="cuda"); e = torch.zeros((n * F, 512), device="cuda")
        t = fa.pipeline_run_dev(det, rec, batches[k].data_ptr(), n, 640, 640, F, f.data_ptr(), o.data_ptr(), e.data_ptr())
        torch.cuda.synchronize()
        ref.append((t, f[:t].clone(), o[:t].clone(), e[:t].clone()))
    sd, sr = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(2):
        for k in range(K):
            f = torch.zeros((n * F, 15), device="cuda"); o = torch.zeros(n * F, dtype=torch.int32, device="cuda")
            e = torch.zeros((n * F, 512), device="cuda"); t = torch.zeros(1, dtype=torch.int32, device="cuda")
            got = fa.pipeline_submit_dev(det, rec, batches[k].data_ptr(), n, 640, 640, F, f.data_ptr(), o.data_ptr(), e.data_ptr(), t.data_ptr(),
                                         sd.cuda_stream, sr.cuda_stream)
            outs.append((k, got, f, o, e))
    torch.cuda.synchronize()
    for k, got, f, o, e in outs:
        rt, rf, ro, re_ = ref[k]
        assert got == rt == n * F
        assert torch.equal(f[:rt].view(torch.int32), rf.view(torch.int32)) and torch.equal(o[:rt], ro) and torch.equal(e[:rt], re_)


def test_split_bf16_mode_is_gated_and_within_tolerance(monkeypatch):
    """Opt-in fh_rec_set_precision(FH_PREC_BF16X2) on the full-size IResNet-50 at the headline batch: the library's gate passes, the
    embeddings stay within the north-star 1 - cos < 1e-3 of BOTH the fp32 path (all 128 slots) and the oracle's direct fp32 evaluation
    (three slots), switching back restores the fp32 bits, and a gate the mode cannot meet leaves the handle fp32."""
    from facerecognizeonnx_amd.synth import models
    path = models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    B = 128
    crops = util.frames_u8(B, 112, 112, seed=4242)
    d = dev(crops)

    def embed():
        out = torch.zeros((B, 512), device="cuda")
        assert rec.embed_aligned_dev(d.data_ptr(), B, out.data_ptr(), 0) == B
        torch.cuda.synchronize()
        return out.cpu().numpy()

    f32 = embed()
    assert rec.precision() == "fp32"
    worst = rec.set_precision("bf16x2")
    assert rec.precision() == "bf16x2" and 0.0 < worst < 1e-3
    b2 = embed()
    assert np.isfinite(b2).all() and not np.array_equal(b2, f32)           # the mode really ran something else
    err = 1.0 - np.sum(b2.astype(np.float64) * f32.astype(np.float64), 1)
    assert err.max() < 1e-3, err.max()
    for i in (0, B // 2, B - 1):
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: oracle.rec_preprocess(crops[i])[None]})[orec.g.outputs[0][0]].reshape(-1)
        ref = oracle.l2_normalize(r)
        assert 1.0 - float(np.dot(b2[i].astype(np.float64), ref.astype(np.float64))) < 1e-3
    print(f"\n[bf16x2] gate batch max(1-cos) = {worst:.3e}; headline batch vs fp32 path = {err.max():.3e}, max |d| = {np.abs(b2 - f32).max():.3e}")
    rec.set_precision("fp32")
    assert rec.precision() == "fp32" and np.array_equal(embed(), f32)
    monkeypatch.setenv("FACEHIP_PRECISION_GATE", "1e-12")                   # a bar the mode cannot meet: refused, handle stays fp32
    with pytest.raises(RuntimeError, match="staying fp32"):
        rec.set_precision("bf16x2")
    assert rec.precision() == "fp32" and np.array_equal(embed(), f32)


def test_split_bf16_mode_refused_for_a_model_without_eligible_layers(models_dir):
    rec = fa.FaceRecognizer()
    assert rec.loadModel(util.tiny_iresnet(models_dir, fold_bn=True))      # widest layer 64 channels: nothing runs as a Winograd GEMM
    with pytest.raises(RuntimeError, match="no layer"):
        rec.set_precision("bf16x2")
    assert rec.precision() == "fp32"


@pytest.mark.parametrize("ds_first", [False, True])
def test_folded_shortcut_matches_oracle_and_the_two_conv_form(tmp_path, ds_first):
    """A strided IResNet block's 1x1 shortcut as a tenth tap of the 3x3 convolution it is added to (engine: POp::sc_src,
    conv_mfma.hip tap_setup) — raw outputs against the oracle and against the library's own two-convolution form, with the shortcut
    written before and after the block's convolutions (arena liveness of the shortcut's input)."""
    from facerecognizeonnx_amd.synth import models
    path = models.make_iresnet(str(tmp_path / f"sc_{int(ds_first)}.onnx"), (1, 2, 1, 1), (32, 64, 128, 128), 112, 64, seed=9, downsample_first=ds_first)
    assert fa.plan_describe(path, 112, 112).count("sc<-op") == 4
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    n = 5                                                                # 5 x 56 x 56 rows: partial last tile + stream-K remainder
    crops = util.frames_u8(n, 112, 112, seed=31)
    got = {}
    for fold in (1, 0):
        assert fa.lib().fh_rec_set_shortcut_fold(rec.handle, fold) == 0
        out = torch.zeros((n, 64), device="cuda"); raw = torch.zeros((n, 64), device="cuda")
        assert rec.embed_aligned_dev(dev(crops).data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
        torch.cuda.synchronize()
        got[fold] = raw.cpu().numpy()
    assert not np.array_equal(got[0], got[1])                             # different summation order: really two code paths
    for i in range(n):
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: oracle.rec_preprocess(crops[i])[None]})[orec.g.outputs[0][0]].reshape(-1)
        for fold in (1, 0):
            np.testing.assert_allclose(got[fold][i], r, rtol=1e-4, atol=1e-4 * np.abs(r).max(), err_msg=f"fold={fold} slot {i}")


@pytest.mark.parametrize("planes0,rows,cols,pitch", [(48, 90, 128, 384), (64, 128, 101, 101 * 3 + 2)])
def test_matrix_core_stem_stride2_letterbox_and_odd_pitch(tmp_path, planes0, rows, cols, pitch):
    """stem_mfma_kernel in the shapes no full-size model reaches: stride 2, 48 / 64 output channels (3 / 4 MFMA channel blocks), a
    letterboxed frame and a frame whose row pitch is not a multiple of 4 (per-row misalignment of the staged window) — against the
    separate preprocess + convolution form and against the oracle's heads."""
    from facerecognizeonnx_amd.synth import models
    path = models.make_scrfd(str(tmp_path / f"s{planes0}.onnx"), (1, 2, 1, 2), (planes0, 8, 16, 24, 32, 48), 8, 16, seed=21, cls_bias=-2.0,
                             static_hw=128)
    assert f"0 CONV k3s2 128x128x4 -> 64x64x{planes0}" in fa.plan_describe(path, 128, 128)
    det = fa.FaceDetector(); odet = oracle.OracleDetector()
    assert det.loadModel(path) and odet.loadModel(path)
    img = np.zeros((2, rows, pitch), np.uint8)
    img[:, :, :cols * 3] = util.frames_u8(2, rows, cols, seed=planes0).reshape(2, rows, cols * 3)
    d = dev(img)
    outs = []
    for fused in (1, 0):
        assert fa.lib().fh_det_set_fused_stem(det.handle, fused) == 0
        assert fa.lib().fh_det_run_network_dev(det.handle, d.data_ptr(), 2, rows, cols, pitch, rows * pitch, 0) == 2
        torch.cuda.synchronize()
        outs.append([o.copy() for o in _det_outputs(det, 2)])
    for a, b in zip(*outs):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)
    frame = np.ascontiguousarray(img[1, :, :cols * 3].reshape(rows, cols, 3))
    inp, _ = oracle.det_preprocess(frame, 128, 128)
    ref = odet.run_network(inp)
    assert len(ref) == len(outs[0])
    for a, r in zip(outs[0], ref):
        np.testing.assert_allclose(a[1].reshape(-1), np.asarray(r).reshape(-1), rtol=1e-4, atol=1e-4)
