"""Host-side mirror of the reference's class API over the C ABI (libfacehip.so).

`FaceDetector` / `FaceRecognizer` keep the reference's method names, argument order, defaults and
error behaviour (reference src/face_detector.h:14-43, src/face_recognizer.h:9-38):
``loadModel`` returns False on failure, ``detect`` / ``extractFeature`` return empty results
instead of raising, ``compareFaces`` returns 0.0 on size mismatch.  Images are numpy
``uint8[rows, cols, 3]`` BGR arrays (the cv::Mat the reference takes).

The ``*_dev`` methods are the batch additions: they take device pointers (e.g.
``torch.Tensor.data_ptr()``) so frames stay resident in HBM between detect and embed.
There is no CPU path here: without libfacehip.so and a GPU these classes fail loudly.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import FACE_DTYPE, check


@dataclass
class FaceBox:
    """struct FaceBox (src/face_detector.h:8-12): box = (x, y, width, height)."""
    box: tuple = (0, 0, 0, 0)
    score: float = 0.0
    landmarks: np.ndarray = field(default_factory=lambda: np.zeros((5, 2), np.float32))

    def to_record(self) -> np.ndarray:
        r = np.zeros(1, FACE_DTYPE)
        r[0]["x"], r[0]["y"], r[0]["w"], r[0]["h"] = self.box
        r[0]["score"] = self.score
        r[0]["lm"] = np.asarray(self.landmarks, np.float32).reshape(10)
        return r

    @staticmethod
    def from_record(r) -> "FaceBox":
        return FaceBox((int(r["x"]), int(r["y"]), int(r["w"]), int(r["h"])), float(r["score"]),
                       np.array(r["lm"], np.float32).reshape(5, 2))


def _img(image) -> Optional[np.ndarray]:
    if image is None:
        return None
    a = np.asarray(image)
    if a.size == 0:
        return None
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise TypeError("image must be uint8[rows, cols, 3] (BGR)")
    if a.strides[2] != 1 or a.strides[1] != 3:
        a = np.ascontiguousarray(a)
    return a


class FaceDetector:
    def __init__(self):
        self._h = None

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "_LIB", None) is not None:
            _lib._LIB.fh_det_destroy(self._h)
        self._h = None

    # -- reference API ---------------------------------------------------------------------
    def loadModel(self, modelPath: str) -> bool:                      # face_detector.cpp:20-90
        self.close()
        self._h = _lib.lib().fh_det_create(str(modelPath).encode())
        return bool(self._h)

    def detect_records(self, image, scoreThreshold: float = 0.5, nmsThreshold: float = 0.4,
                       max_faces: Optional[int] = None) -> np.ndarray:
        if not self._h:
            return np.zeros(0, FACE_DTYPE)                            # "Model not loaded!" :142-145
        a = _img(image)
        if a is None:
            return np.zeros(0, FACE_DTYPE)                            # :148-156
        if max_faces is None:                                         # the reference's std::vector is unbounded: every candidate row can survive
            max_faces = max(1, self.num_anchors())
        out = np.zeros(max_faces, FACE_DTYPE)
        n = check(_lib.lib().fh_det_detect(self._h, a.ctypes.data, a.shape[0], a.shape[1], a.strides[0],
                                            scoreThreshold, nmsThreshold, out.ctypes.data, max_faces), "fh_det_detect")
        return out[:n].copy()

    def detect(self, image, scoreThreshold: float = 0.5, nmsThreshold: float = 0.4) -> List[FaceBox]:
        return [FaceBox.from_record(r) for r in self.detect_records(image, scoreThreshold, nmsThreshold)]

    # -- batch / device additions -----------------------------------------------------------
    @property
    def handle(self):
        return self._h

    def input_size(self):
        w, h = C.c_int(), C.c_int()
        check(_lib.lib().fh_det_input_size(self._h, C.byref(w), C.byref(h)), "fh_det_input_size")
        return w.value, h.value

    def num_anchors(self) -> int:
        return _lib.lib().fh_det_num_anchors(self._h)

    def macs_per_frame(self) -> float:
        return _lib.lib().fh_det_macs_per_frame(self._h)

    def detect_batch_dev(self, frames_ptr: int, n: int, rows: int, cols: int, out_ptr: int, max_per_frame: int,
                         counts_ptr: int, scoreThreshold: float = 0.5, nmsThreshold: float = 0.4,
                         step: int = 0, frame_stride: int = 0, stream: int = 0) -> int:
        step = step or cols * 3
        frame_stride = frame_stride or rows * step
        return check(_lib.lib().fh_det_detect_batch_dev(self._h, frames_ptr, n, rows, cols, step, frame_stride,
                                                        scoreThreshold, nmsThreshold, out_ptr, max_per_frame,
                                                        counts_ptr, stream), "fh_det_detect_batch_dev")

    def sync(self, stream: int = 0) -> None:
        """Waits for `stream`; raises if a launch of this handle failed after its asynchronous call returned (fh_det_sync)."""
        check(_lib.lib().fh_det_sync(self._h, stream), "fh_det_sync")


class FaceRecognizer:
    def __init__(self):
        self._h = None

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "_LIB", None) is not None:
            _lib._LIB.fh_rec_destroy(self._h)
        self._h = None

    # -- reference API ---------------------------------------------------------------------
    def loadModel(self, modelPath: str) -> bool:                      # face_recognizer.cpp:21-91
        self.close()
        self._h = _lib.lib().fh_rec_create(str(modelPath).encode())
        return bool(self._h)

    def extractFeature(self, image, face) -> np.ndarray:             # face_recognizer.cpp:236-304
        if not self._h:
            return np.zeros(0, np.float32)
        a = _img(image)
        if a is None:
            return np.zeros(0, np.float32)
        rec = face.to_record() if isinstance(face, FaceBox) else np.asarray(face, FACE_DTYPE).reshape(1)
        dim = self.feature_dim()
        out = np.zeros(dim, np.float32)
        n = check(_lib.lib().fh_rec_extract(self._h, a.ctypes.data, a.shape[0], a.shape[1], a.strides[0],
                                             rec.ctypes.data, out.ctypes.data, dim), "fh_rec_extract")
        return out[:n]

    def extractFeatureSimple(self, image) -> np.ndarray:             # face_recognizer.cpp:152-234
        if not self._h:
            return np.zeros(0, np.float32)
        a = _img(image)
        if a is None:
            return np.zeros(0, np.float32)
        dim = self.feature_dim()
        out = np.zeros(dim, np.float32)
        n = check(_lib.lib().fh_rec_extract_simple(self._h, a.ctypes.data, a.shape[0], a.shape[1], a.strides[0],
                                                    out.ctypes.data, dim), "fh_rec_extract_simple")
        return out[:n]

    @staticmethod
    def compareFaces(feature1, feature2) -> float:                   # face_recognizer.cpp:320-334
        f1 = np.ascontiguousarray(feature1, np.float32).reshape(-1)
        f2 = np.ascontiguousarray(feature2, np.float32).reshape(-1)
        return float(_lib.lib().fh_compare(f1.ctypes.data, f1.size, f2.ctypes.data, f2.size))

    # -- batch / device additions -----------------------------------------------------------
    @property
    def handle(self):
        return self._h

    def feature_dim(self) -> int:
        return _lib.lib().fh_rec_feature_dim(self._h)

    def macs_per_face(self) -> float:
        return _lib.lib().fh_rec_macs_per_face(self._h)

    def set_chunk(self, n: int):
        check(_lib.lib().fh_rec_set_chunk(self._h, n), "fh_rec_set_chunk")

    def set_precision(self, mode: str = "fp32") -> float:
        """"fp32" (default, the reference's arithmetic) or "bf16x2" (split-bf16 Winograd GEMMs, opt-in).  The library gates the
        switch on its own check (max 1 - cos < 1e-3 against fp32 on a fixed batch) and raises if it fails, staying fp32.
        Returns the measured max(1 - cos)."""
        import ctypes as C
        modes = {"fp32": 0, "bf16x2": 1}
        if mode not in modes:
            raise ValueError(f"precision {mode!r}: expected one of {sorted(modes)}")
        worst = C.c_float(0.0)
        check(_lib.lib().fh_rec_set_precision(self._h, modes[mode], C.byref(worst)), "fh_rec_set_precision")
        return float(worst.value)

    def precision(self) -> str:
        return "bf16x2" if _lib.lib().fh_rec_get_precision(self._h) == 1 else "fp32"

    def embed_aligned_dev(self, crops_ptr: int, n: int, out_ptr: int, raw_ptr: int = 0, stream: int = 0) -> int:
        return check(_lib.lib().fh_rec_embed_aligned_dev(self._h, crops_ptr, n, out_ptr, raw_ptr, stream),
                     "fh_rec_embed_aligned_dev")

    def sync(self, stream: int = 0) -> None:
        """Waits for `stream`; raises if a launch of this handle failed after its asynchronous call returned (fh_rec_sync)."""
        check(_lib.lib().fh_rec_sync(self._h, stream), "fh_rec_sync")


def pipeline_run_dev(det: FaceDetector, rec: FaceRecognizer, frames_ptr: int, n: int, rows: int, cols: int,
                     faces_per_frame: int, faces_ptr: int, frame_of_ptr: int, emb_ptr: int,
                     scoreThreshold: float = 0.5, nmsThreshold: float = 0.4, stream: int = 0) -> int:
    """detect -> align -> embed on n HBM-resident frames; returns the number of faces embedded."""
    step = cols * 3
    return check(_lib.lib().fh_pipeline_run_dev(det.handle, rec.handle, frames_ptr, n, rows, cols, step, rows * step,
                                                scoreThreshold, nmsThreshold, faces_per_frame, faces_ptr,
                                                frame_of_ptr, emb_ptr, stream), "fh_pipeline_run_dev")


def pipeline_submit_dev(det: FaceDetector, rec: FaceRecognizer, frames_ptr: int, n: int, rows: int, cols: int,
                        faces_per_frame: int, faces_ptr: int, frame_of_ptr: int, emb_ptr: int, total_ptr: int,
                        stream_det: int, stream_rec: int, scoreThreshold: float = 0.5, nmsThreshold: float = 0.4) -> int:
    """Two-stream detect -> align -> embed: detector on stream_det, recogniser on stream_rec; the host waits for the
    detector's face count only.  Returns the number of faces."""
    step = cols * 3
    return check(_lib.lib().fh_pipeline_submit_dev(det.handle, rec.handle, frames_ptr, n, rows, cols, step, rows * step,
                                                   scoreThreshold, nmsThreshold, faces_per_frame, faces_ptr, frame_of_ptr,
                                                   emb_ptr, total_ptr, stream_det, stream_rec), "fh_pipeline_submit_dev")


class FrameStream:
    """Streaming front end for batches of HOST frames (fh_stream_*): the reference's webcam loop (src/main.cpp:214-258) over
    batches.  submit() uploads + queues a batch and returns its face count, collect() returns the oldest batch's results."""

    def __init__(self, det: FaceDetector, rec: FaceRecognizer, frames_per_batch: int, rows: int, cols: int, faces_per_frame: int = 1):
        self._h = _lib.lib().fh_stream_create(det.handle, rec.handle, frames_per_batch, rows, cols, faces_per_frame)
        if not self._h:
            raise _lib.FaceHipError("fh_stream_create failed: " + _lib.last_error())
        self._keep = (det, rec)
        self.cap = frames_per_batch * faces_per_frame
        self.dim = rec.feature_dim()

    def close(self):
        if getattr(self, "_h", None) and getattr(_lib, "_LIB", None) is not None:
            _lib._LIB.fh_stream_destroy(self._h)
        self._h = None

    __del__ = close

    def submit(self, frames, scoreThreshold: float = 0.5, nmsThreshold: float = 0.4) -> int:
        """frames: contiguous uint8 [n, rows, cols, 3] BGR (a numpy array, or an int pointer with n given as frames[1])."""
        if isinstance(frames, tuple):
            ptr, n = frames
        else:
            a = np.ascontiguousarray(frames, np.uint8)
            ptr, n = a.ctypes.data, a.shape[0]
        return check(_lib.lib().fh_stream_submit(self._h, ptr, n, scoreThreshold, nmsThreshold), "fh_stream_submit")

    def collect(self):
        faces = np.zeros(self.cap, FACE_DTYPE); frame_of = np.zeros(self.cap, np.int32); emb = np.zeros((self.cap, self.dim), np.float32)
        n = check(_lib.lib().fh_stream_collect(self._h, faces.ctypes.data, frame_of.ctypes.data, emb.ctypes.data, self.cap), "fh_stream_collect")
        return faces[:n], frame_of[:n], emb[:n]

    def collect_count(self) -> int:
        return check(_lib.lib().fh_stream_collect(self._h, None, None, None, 0), "fh_stream_collect")


class Gallery:
    """1:N generalisation of compareFaces: top-k mapped scores (dot+1)/2 over enrolled rows."""

    def __init__(self, dim: int = 512):
        self._h = _lib.lib().fh_gallery_create(dim)
        self.dim = dim

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and getattr(_lib, "_LIB", None) is not None:
            _lib._LIB.fh_gallery_destroy(self._h)
        self._h = None

    def upload(self, rows_ptr: int, n: int, on_device: bool, index_base: int = 0):
        check(_lib.lib().fh_gallery_upload(self._h, rows_ptr, n, int(on_device), index_base), "fh_gallery_upload")

    def topk_dev(self, q_ptr: int, nq: int, k: int, scores_ptr: int, idx_ptr: int, stream: int = 0):
        check(_lib.lib().fh_gallery_topk_dev(self._h, q_ptr, nq, k, scores_ptr, idx_ptr, stream), "fh_gallery_topk_dev")

    def enroll(self, rows) -> int:
        """Append L2-normalised feature rows (host array [n, dim]); the webcam loop's 's' key (main.cpp:253-256).
        Returns the index of the first new row."""
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, self.dim)
        return check(_lib.lib().fh_gallery_enroll(self._h, rows.ctypes.data, rows.shape[0], 0), "fh_gallery_enroll")

    def __len__(self) -> int:
        return int(_lib.lib().fh_gallery_size(self._h))

    def label_dev(self, q_ptr: int, nq: int, threshold: float, labels_ptr: int, scores_ptr: int, stream: int = 0):
        """labels[q] = best row if (dot+1)/2 > threshold else -1 ("Match" / "Unknown", main.cpp:229-233)."""
        check(_lib.lib().fh_gallery_label_dev(self._h, q_ptr, nq, threshold, labels_ptr, scores_ptr, stream), "fh_gallery_label_dev")


class Comm:
    """RCCL communicator behind the C ABI (fh_comm_*): one per process / rank.  `Comm.unique_id()` on rank 0, hand the 128 bytes to
    the other ranks (torch.distributed's store, MPI, a file), then `Comm(rank, world, id, device)` on every rank."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * 128)()
        check(_lib.lib().fh_comm_unique_id(C.cast(buf, C.c_void_p)), "fh_comm_unique_id")
        return bytes(buf)

    def __init__(self, rank: int, world: int, uid: bytes, device: int = 0):
        if len(uid) != 128:
            raise ValueError("unique id must be 128 bytes")
        self._id = (C.c_ubyte * 128).from_buffer_copy(uid)
        self._h = _lib.lib().fh_comm_create(rank, world, C.cast(self._id, C.c_void_p), device)
        if not self._h:
            raise _lib.FaceHipError("fh_comm_create failed: " + _lib.last_error())
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().fh_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._h

    def allgather_f32_dev(self, send_ptr: int, recv_ptr: int, count: int, stream: int = 0) -> int:
        return check(_lib.lib().fh_comm_allgather_f32_dev(self._h, send_ptr, recv_ptr, count, stream), "fh_comm_allgather_f32_dev")

    def gallery_topk_sharded_dev(self, gallery: "Gallery", q_ptr: int, nq_local: int, k: int, scores_ptr: int, idx_ptr: int,
                                 stream: int = 0) -> int:
        """queries all-gather -> local shard scan -> one top-k all-gather -> merge, all on `stream` (fh_gallery_topk_sharded_dev);
        scores / indices = [world * nq_local][k] on every rank."""
        return check(_lib.lib().fh_gallery_topk_sharded_dev(gallery._h, self._h, q_ptr, nq_local, k, scores_ptr, idx_ptr, stream),
                     "fh_gallery_topk_sharded_dev")


def plan_describe(path: str, default_h: int, default_w: int) -> str:
    buf = C.create_string_buffer(1 << 18)
    check(_lib.lib().fh_plan_describe(str(path).encode(), default_h, default_w, buf, len(buf)), "fh_plan_describe")
    return buf.value.decode()


def imread(path: str) -> Optional[np.ndarray]:
    """cv::imread(path) (reference src/main.cpp:42): BGR u8 [rows, cols, 3], or None when the file cannot be read or
    decoded (cv::imread returns an empty Mat).  JPEG / PNG / BMP / PPM, decoded by the library's own host code."""
    L = _lib.lib()
    p, r, c = C.c_void_p(), C.c_int(), C.c_int()
    if L.fh_imread(str(path).encode(), C.byref(p), C.byref(r), C.byref(c)) != 0:
        return None
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), (r.value, c.value, 3)).copy()
    finally:
        L.fh_image_free(p)
