"""CPU ORACLE driver — test infrastructure only (see face_oracle.c header).

PARITY WITH ORT / OPENCV IS UNPINNED: neither library nor a model file exists offline and
the reference has no golden vectors.  This restatement is pinned by
hand-derived KATs and by an independent PyTorch-CPU fp64 evaluation (oracle/torch_graph.py).

`OracleDetector` / `OracleRecognizer` restate reference `FaceDetector` / `FaceRecognizer`
(src/face_detector.cpp, src/face_recognizer.cpp): same method names, same error behaviour
(empty results), batch = 1, node-by-node unfused NCHW graph evaluation as ONNX Runtime's
CPU provider would do it.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import onnx_min

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TEMPLATE = np.array([38.2946, 51.6963, 73.5318, 51.5014, 56.0252, 71.7366,
                     41.5493, 92.3655, 70.7299, 92.2041], np.float32)   # face_recognizer.cpp:101-107

FACE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"),
                       ("score", "<f4"), ("lm", "<f4", (10,))])
assert FACE_DTYPE.itemsize == 60


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libface_oracle.so")
    src = os.path.join(_HERE, "face_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libface_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_iou.restype = C.c_float
        _LIB.orc_compare.restype = C.c_float
    return _LIB


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def set_threads(n: int):
    lib().orc_set_threads(int(n))


# ------------------------------------------------------------------ graph operators (NCHW)
def conv2d(x, w, b, stride, pad, group):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    y = np.empty((N, Cout, Ho, Wo), np.float32)
    bp = _f(np.ascontiguousarray(b, np.float32)) if b is not None else None
    lib().orc_conv2d(_f(x), N, Cin, H, W, _f(w), bp, Cout, kh, kw, stride, pad, group, _f(y))
    return y


def batchnorm(x, g, b, m, v, eps):
    x = np.ascontiguousarray(x, np.float32)
    shp = x.shape
    N, Cc = shp[0], shp[1]
    HW = int(np.prod(shp[2:])) if len(shp) > 2 else 1
    y = np.empty_like(x)
    lib().orc_batchnorm(_f(x), N, Cc, HW, _f(g), _f(b), _f(m), _f(v), C.c_float(eps), _f(y))
    return y


def prelu(x, slope):
    x = np.ascontiguousarray(x, np.float32)
    N, Cc = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    s = np.ascontiguousarray(slope, np.float32).reshape(-1)
    if s.size == 1:
        s = np.repeat(s, Cc)
    y = np.empty_like(x)
    lib().orc_prelu(_f(x), N, Cc, HW, _f(s), _f(y))
    return y


def _ew(fn, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    getattr(lib(), fn)(_f(x), C.c_size_t(x.size), _f(y))
    return y


def run_graph(g: onnx_min.Graph, feeds: dict) -> dict:
    """Evaluate every node in file order, one kernel per node (no fusion)."""
    env = dict(g.inits)
    env.update({k: np.ascontiguousarray(v, np.float32) for k, v in feeds.items()})
    L = lib()
    for n in g.nodes:
        a = n.attrs
        i = [env[k] if k else None for k in n.inputs]
        if n.op == "Conv":
            k = a.get("kernel_shape", list(i[1].shape[2:]))
            st = a.get("strides", [1, 1]); pd = a.get("pads", [0, 0, 0, 0])
            assert k[0] == k[1] and st[0] == st[1] and len(set(pd)) == 1 and a.get("dilations", [1, 1]) == [1, 1]
            y = conv2d(i[0], i[1], i[2] if len(i) > 2 else None, st[0], pd[0], a.get("group", 1))
        elif n.op == "BatchNormalization":
            y = batchnorm(i[0], i[1], i[2], i[3], i[4], a.get("epsilon", 1e-5))
        elif n.op == "PRelu":
            y = prelu(i[0], i[1])
        elif n.op == "Relu":
            y = _ew("orc_relu", i[0])
        elif n.op == "Sigmoid":
            y = _ew("orc_sigmoid", i[0])
        elif n.op in ("Identity", "Dropout"):                   # inference: pass-through
            y = i[0]
        elif n.op in ("Mul", "Sub", "Div") or (n.op == "Add" and np.shape(i[0]) != np.shape(i[1])):
            # element-wise op against a constant (scalar / per-channel, numpy broadcasting), evaluated literally in fp32 as ORT does
            f = {"Mul": np.multiply, "Sub": np.subtract, "Div": np.divide, "Add": np.add}[n.op]
            y = f(np.asarray(i[0], np.float32), np.asarray(i[1], np.float32), dtype=np.float32)
        elif n.op == "Add":
            x0 = np.ascontiguousarray(i[0], np.float32); x1 = np.ascontiguousarray(i[1], np.float32)
            assert x0.shape == x1.shape
            y = np.empty_like(x0)
            L.orc_add(_f(x0), _f(x1), C.c_size_t(x0.size), _f(y))
        elif n.op in ("Shape", "Slice", "Concat", "Cast", "Constant", "Gather", "Unsqueeze"):
            # small integer shape arithmetic of dynamic-axes exports (evaluated literally, as ORT does)
            if n.op == "Shape":
                y = np.array(i[0].shape, np.int64)
            elif n.op == "Slice":
                st, en = int(i[1][0]), int(i[2][0])
                y = np.asarray(i[0])[st:en]
            elif n.op == "Concat":
                y = np.concatenate([np.atleast_1d(v) for v in i])
            elif n.op == "Cast":
                y = np.asarray(i[0]).astype(np.int64 if a.get("to", 7) == 7 else np.float32)
            elif n.op == "Constant":
                y = a["value"]
            elif n.op == "Gather":
                y = np.asarray(i[0])[np.asarray(i[1], np.int64)]
            else:
                y = np.atleast_1d(i[0])
        elif n.op == "Resize":
            scales = i[2] if len(i) > 2 and i[2] is not None and i[2].size else None
            if scales is not None:
                s = int(scales[2]); assert list(scales) == [1, 1, s, s]
            else:
                s = int(i[3][2]) // i[0].shape[2]
            assert a.get("mode", "nearest") == "nearest"
            x0 = np.ascontiguousarray(i[0], np.float32)
            N, Cc, H, W = x0.shape
            y = np.empty((N, Cc, H * s, W * s), np.float32)
            L.orc_resize_nearest(_f(x0), N * Cc, H, W, s, _f(y))
        elif n.op == "Transpose":
            assert a["perm"] == [0, 2, 3, 1]
            x0 = np.ascontiguousarray(i[0], np.float32)
            N, Cc, H, W = x0.shape
            y = np.empty((N, H, W, Cc), np.float32)
            L.orc_nchw_to_nhwc(_f(x0), N, Cc, H, W, _f(y))
        elif n.op == "Reshape":
            y = i[0].reshape([int(d) for d in i[1]])
        elif n.op == "Flatten":
            y = i[0].reshape(i[0].shape[0], -1)
        elif n.op == "MatMul":                                  # [M,K] x constant [K,N] (a bias-less Linear)
            x0 = np.ascontiguousarray(i[0], np.float32)
            wt = np.ascontiguousarray(np.asarray(i[1], np.float32).T)
            y = np.empty((x0.shape[0], wt.shape[0]), np.float32)
            L.orc_gemm_nt(_f(x0), x0.shape[0], x0.shape[1], _f(wt), None, wt.shape[0], _f(y))
        elif n.op == "Gemm":
            assert a.get("transB", 0) == 1 and a.get("alpha", 1.0) == 1.0 and a.get("beta", 1.0) == 1.0
            x0 = np.ascontiguousarray(i[0], np.float32)
            M, K = x0.shape
            Nn = i[1].shape[0]
            y = np.empty((M, Nn), np.float32)
            bp = _f(np.ascontiguousarray(i[2], np.float32)) if len(i) > 2 else None
            L.orc_gemm_nt(_f(x0), M, K, _f(np.ascontiguousarray(i[1], np.float32)), bp, Nn, _f(y))
        else:
            raise NotImplementedError(n.op)
        env[n.outputs[0]] = y
    return {name: env[name] for name, _ in g.outputs}


# ------------------------------------------------------------------ pipeline pieces
def det_preprocess(img: np.ndarray, inW: int, inH: int):
    """FaceDetector::preprocess (face_detector.cpp:92-137) → ([3,inH,inW] fp32 | None, scale)."""
    scale = C.c_float(1.0)
    if img is None or img.size == 0:
        return None, 1.0
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((3, inH, inW), np.float32)
    ok = lib().orc_det_preprocess(_u8(img), img.shape[0], img.shape[1], img.strides[0],
                                  inW, inH, _f(out), C.byref(scale))
    return (out if ok else None), scale.value


def scrfd_decode(outs: list, inH: int, inW: int) -> np.ndarray:
    """9 SCRFD outputs (score×3, bbox×3, kps×3) → [N,15] rows."""
    arrs = [np.ascontiguousarray(o, np.float32) for o in outs]
    PP = C.POINTER(C.c_float) * 3
    n = sum(a.shape[0] for a in arrs[:3])
    rows = np.empty((n, 15), np.float32)
    r = lib().orc_scrfd_decode(PP(*[_f(a) for a in arrs[0:3]]), PP(*[_f(a) for a in arrs[3:6]]),
                               PP(*[_f(a) for a in arrs[6:9]]), inH, inW, _f(rows))
    assert r == n
    return rows


def postprocess_rows(rows: np.ndarray, scale: float, score_thr: float, nms_thr: float) -> np.ndarray:
    """postprocess row loop + nms (face_detector.cpp:249-338) → FACE_DTYPE array."""
    rows = np.ascontiguousarray(rows, np.float32).reshape(-1, rows.shape[-1])
    n, feat = rows.shape
    faces = np.zeros(max(n, 1), FACE_DTYPE)
    m = lib().orc_postprocess_rows(_f(rows), n, feat, C.c_float(scale), C.c_float(score_thr),
                                   faces.ctypes.data_as(C.c_void_p), n)
    m = lib().orc_nms(faces.ctypes.data_as(C.c_void_p), m, C.c_float(nms_thr))
    return faces[:m].copy()


def threshold_rows(rows, scale, score_thr):
    rows = np.ascontiguousarray(rows, np.float32).reshape(-1, rows.shape[-1])
    n, feat = rows.shape
    faces = np.zeros(max(n, 1), FACE_DTYPE)
    m = lib().orc_postprocess_rows(_f(rows), n, feat, C.c_float(scale), C.c_float(score_thr),
                                   faces.ctypes.data_as(C.c_void_p), n)
    return faces[:m].copy()


def nms(faces: np.ndarray, thr: float) -> np.ndarray:
    f = np.ascontiguousarray(faces.copy())
    m = lib().orc_nms(f.ctypes.data_as(C.c_void_p), len(f), C.c_float(thr))
    return f[:m].copy()


def iou(a, b) -> float:
    fa = np.zeros(1, FACE_DTYPE); fb = np.zeros(1, FACE_DTYPE)
    fa[0]["x"], fa[0]["y"], fa[0]["w"], fa[0]["h"] = a
    fb[0]["x"], fb[0]["y"], fb[0]["w"], fb[0]["h"] = b
    return float(lib().orc_iou(fa.ctypes.data_as(C.c_void_p), fb.ctypes.data_as(C.c_void_p)))


def estimate_similarity(src5, dst5=TEMPLATE):
    M = np.zeros(6, np.float64)
    s = np.ascontiguousarray(src5, np.float32).reshape(-1)
    d = np.ascontiguousarray(dst5, np.float32).reshape(-1)
    ok = lib().orc_estimate_similarity5(_f(s), _f(d), M.ctypes.data_as(C.POINTER(C.c_double)))
    return M.reshape(2, 3) if ok else None


def warp_affine(img, M, dw=112, dh=112):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dh, dw, 3), np.uint8)
    Mc = np.ascontiguousarray(M, np.float64).reshape(-1)
    lib().orc_warp_affine_u8c3(_u8(img), img.shape[0], img.shape[1], img.strides[0],
                               Mc.ctypes.data_as(C.POINTER(C.c_double)), _u8(out), dh, dw, dw * 3)
    return out


def resize_bilinear(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((dh, dw, 3), np.uint8)
    lib().orc_resize_bilinear_u8c3(_u8(img), img.shape[0], img.shape[1], img.strides[0], _u8(out), dh, dw, dw * 3)
    return out


def align_face(img, face, outW=112, outH=112):
    """FaceRecognizer::alignFace (face_recognizer.cpp:93-133) → [outH,outW,3] u8 | None."""
    if img is None or img.size == 0:
        return None
    img = np.ascontiguousarray(img, np.uint8)
    f = np.zeros(1, FACE_DTYPE); f[0] = face
    out = np.empty((outH, outW, 3), np.uint8)
    ok = lib().orc_align_face(_u8(img), img.shape[0], img.shape[1], img.strides[0],
                              f.ctypes.data_as(C.c_void_p), outW, outH, _u8(out))
    return out if ok else None


def rec_preprocess(aligned):
    a = np.ascontiguousarray(aligned, np.uint8)
    out = np.empty((3, a.shape[0], a.shape[1]), np.float32)
    lib().orc_rec_preprocess(_u8(a), a.shape[0], a.shape[1], _f(out))
    return out


def l2_normalize(v):
    v = np.ascontiguousarray(v, np.float32).copy().reshape(-1)
    lib().orc_l2_normalize(_f(v), v.size)
    return v


def compare(a, b) -> float:
    a = np.ascontiguousarray(a, np.float32).reshape(-1); b = np.ascontiguousarray(b, np.float32).reshape(-1)
    return float(lib().orc_compare(_f(a), a.size, _f(b), b.size))


def gallery_topk(q, gal, k):
    q = np.ascontiguousarray(q, np.float32); gal = np.ascontiguousarray(gal, np.float32)
    Q, dim = q.shape; G = gal.shape[0]
    s = np.empty((Q, k), np.float32); i = np.empty((Q, k), np.int32)
    lib().orc_gallery_topk(_f(q), Q, _f(gal), G, dim, k, _f(s), i.ctypes.data_as(C.POINTER(C.c_int)))
    return s, i


# ------------------------------------------------------------------ reference-shaped classes
class OracleDetector:
    """Restates FaceDetector (face_detector.h:14-43)."""

    def __init__(self):
        self.g = None
        self.inW = self.inH = 640                      # face_detector.cpp:8-9

    def loadModel(self, path: str) -> bool:            # face_detector.cpp:20-90
        try:
            self.g = onnx_min.load(path)
        except Exception:
            return False
        shp = self.g.inputs[0][1]
        if len(shp) == 4:
            if shp[2] > 0:
                self.inH = int(shp[2])
            if shp[3] > 0:
                self.inW = int(shp[3])
        return True

    def run_network(self, inp):
        outs = run_graph(self.g, {self.g.inputs[0][0]: inp[None]})
        return [outs[n] for n, _ in self.g.outputs]

    def rows_from_outputs(self, outs):
        """Reference looks only at output 0 ([b,N,>=15] or [N,>=15]); a genuine 9-output
        SCRFD graph is decoded first."""
        if len(outs) == 9:
            return scrfd_decode(outs, self.inH, self.inW)
        o = outs[0]
        if o.ndim == 3 and o.shape[2] >= 15:
            return o[0]
        if o.ndim == 2:
            return o
        return np.zeros((0, 15), np.float32)           # "Unexpected output shape" :326-328

    def detect(self, img, scoreThreshold=0.5, nmsThreshold=0.4):
        if self.g is None or img is None or img.size == 0:
            return np.zeros(0, FACE_DTYPE)             # :142-156
        inp, scale = det_preprocess(img, self.inW, self.inH)
        if inp is None:
            return np.zeros(0, FACE_DTYPE)             # :164-167
        rows = self.rows_from_outputs(self.run_network(inp))
        if rows.shape[0] == 0 or rows.shape[1] < 15:
            return np.zeros(0, FACE_DTYPE)
        return postprocess_rows(rows, scale, scoreThreshold, nmsThreshold)


class OracleRecognizer:
    """Restates FaceRecognizer (face_recognizer.h:9-38)."""

    def __init__(self):
        self.g = None
        self.inW = self.inH = 112                      # face_recognizer.cpp:8-9

    def loadModel(self, path: str) -> bool:
        try:
            self.g = onnx_min.load(path)
        except Exception:
            return False
        shp = self.g.inputs[0][1]
        if len(shp) == 4:
            if shp[2] > 0:
                self.inH = int(shp[2])
            if shp[3] > 0:
                self.inW = int(shp[3])
        return True

    def embed_aligned(self, aligned):
        inp = rec_preprocess(aligned)
        out = run_graph(self.g, {self.g.inputs[0][0]: inp[None]})[self.g.outputs[0][0]]
        return l2_normalize(out.reshape(-1))           # :286-297

    def extractFeature(self, img, face):
        if self.g is None or img is None or img.size == 0:
            return np.zeros(0, np.float32)             # :239-248
        aligned = align_face(img, face, self.inW, self.inH)
        if aligned is None:
            return np.zeros(0, np.float32)             # :254-257
        return self.embed_aligned(aligned)

    def extractFeatureSimple(self, img):               # :152-234
        if self.g is None or img is None or img.size == 0:
            return np.zeros(0, np.float32)
        return self.embed_aligned(resize_bilinear(img, self.inW, self.inH))

    @staticmethod
    def compareFaces(f1, f2) -> float:                 # :320-334
        return compare(f1, f2)
