"""ctypes binding of libfacehip.so (the C ABI in include/facehip.h).

The library is the product; there is no CPU fallback.  If it is missing it is built with
`make` (hipcc cross-compiles gfx950 without a GPU); if that fails the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfacehip.so")
CSRC = os.path.join(_HERE, "csrc")

FACE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("w", "<i4"), ("h", "<i4"),
                       ("score", "<f4"), ("lm", "<f4", (10,))])
assert FACE_DTYPE.itemsize == 60


class FhFace(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("score", C.c_float), ("lm", C.c_float * 10)]


def build(force: bool = False) -> str:
    """Compile libfacehip.so in-tree (so that it travels with the source snapshot)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "facehip.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
    return LIB_PATH


def csrc_fingerprint() -> str:
    """sha256 (first 16 hex digits) over the kernel / host sources the library is built from: the stamp that ties a committed counter
    profile (profiles/traffic.json) to the code it was measured on (bench.py reports `roofline.traffic` only when it still matches)."""
    import hashlib
    h = hashlib.sha256()
    host_only = {"image_io.cpp", "onnx_reader.cpp", "onnx_reader.h", "comm.cpp"}     # file parsers / RCCL glue: no kernel, no launch decision
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp", ".h")) and f not in host_only:
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


_vp, _i, _f, _ll, _d = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_double
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); one row per symbol declared in include/facehip.h
PROTOTYPES = {
    "fh_version": (C.c_char_p, []),
    "fh_last_error": (C.c_char_p, []),
    "fh_init": (_i, [_i]),
    "fh_plan_describe": (_i, [C.c_char_p, _i, _i, C.c_char_p, _i]),
    "fh_onnx_dump": (_i, [C.c_char_p, C.c_char_p, _i]),
    "fh_det_create": (_vp, [C.c_char_p]),
    "fh_det_destroy": (None, [_vp]),
    "fh_det_input_size": (_i, [_vp, _ip, _ip]),
    "fh_det_num_anchors": (_i, [_vp]),
    "fh_det_macs_per_frame": (_d, [_vp]),
    "fh_det_act_bytes_per_frame": (_d, [_vp]),
    "fh_det_detect": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp, _i]),
    "fh_det_detect_batch_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _ll, _f, _f, _vp, _i, _vp, _vp]),
    "fh_det_run_network_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _ll, _vp]),
    "fh_det_num_outputs": (_i, [_vp]),
    "fh_det_output_dev": (_vp, [_vp, _i, _ip, _ip]),
    "fh_det_input_dev": (_vp, [_vp]),
    "fh_det_postprocess_dev": (_i, [_vp, _i, _f, _f, _vp, _i, _vp, _vp]),
    "fh_postprocess_rows_dev": (_i, [_vp, _i, _i, _i, _f, _f, _f, _vp, _i, _vp, _vp]),
    "fh_rec_create": (_vp, [C.c_char_p]),
    "fh_rec_destroy": (None, [_vp]),
    "fh_rec_input_size": (_i, [_vp, _ip, _ip]),
    "fh_rec_feature_dim": (_i, [_vp]),
    "fh_rec_macs_per_face": (_d, [_vp]),
    "fh_rec_act_bytes_per_face": (_d, [_vp]),
    "fh_rec_set_chunk": (_i, [_vp, _i]),
    "fh_rec_extract": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i]),
    "fh_rec_extract_simple": (_i, [_vp, _vp, _i, _i, _i, _vp, _i]),
    "fh_compare": (_f, [_vp, _i, _vp, _i]),
    "fh_rec_embed_aligned_dev": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "fh_rec_align_dev": (_i, [_vp, _vp, _i, _i, _i, _ll, _vp, _vp, _i, _vp, _vp, _vp]),
    "fh_rec_input_dev": (_vp, [_vp]),
    "fh_rec_embed_faces_dev": (_i, [_vp, _vp, _i, _i, _i, _ll, _vp, _vp, _i, _vp, _vp, _vp]),
    "fh_pipeline_run_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ll, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "fh_pipeline_submit_dev": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _ll, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fh_stream_create": (_vp, [_vp, _vp, _i, _i, _i, _i]),
    "fh_stream_destroy": (None, [_vp]),
    "fh_stream_submit": (_i, [_vp, _vp, _i, _f, _f]),
    "fh_stream_collect": (_i, [_vp, _vp, _vp, _vp, _i]),
    "fh_gallery_create": (_vp, [_i]),
    "fh_gallery_destroy": (None, [_vp]),
    "fh_gallery_upload": (_i, [_vp, _vp, _ll, _i, _ll]),
    "fh_gallery_enroll": (_ll, [_vp, _vp, _ll, _i]),
    "fh_gallery_size": (_ll, [_vp]),
    "fh_gallery_label_dev": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp]),
    "fh_gallery_topk_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "fh_topk_merge_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "fh_comm_unique_id": (_i, [_vp]),
    "fh_comm_create": (_vp, [_i, _i, _vp, _i]),
    "fh_comm_destroy": (None, [_vp]),
    "fh_comm_rank": (_i, [_vp]),
    "fh_comm_world": (_i, [_vp]),
    "fh_comm_allgather_f32_dev": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "fh_gallery_topk_sharded_dev": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "fh_timing_enable": (_i, [_i]),
    "fh_timing_num_tags": (_i, []),
    "fh_timing_collect": (_i, [_vp, _vp, _vp, _vp, _i]),
    "fh_timing_collect_ops": (_i, [_vp, _vp, _vp, _i]),
    "fh_det_set_conv_cfg": (_i, [_vp, _i, _i]),
    "fh_rec_set_conv_cfg": (_i, [_vp, _i, _i]),
    "fh_memcpy_d2h": (_i, [_vp, _vp, C.c_size_t]),
    "fh_det_sync": (_i, [_vp, _vp]),
    "fh_rec_sync": (_i, [_vp, _vp]),
    "fh_imread": (_i, [C.c_char_p, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)]),
    "fh_image_decode": (_i, [_vp, C.c_size_t, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)]),
    "fh_image_free": (None, [_vp]),
    "fh_det_set_winograd": (_i, [_vp, _i]),
    "fh_rec_set_winograd": (_i, [_vp, _i]),
    "fh_rec_set_wino_fusion": (_i, [_vp, _i]),
    "fh_rec_set_shortcut_fold": (_i, [_vp, _i]),
    "fh_rec_set_precision": (_i, [_vp, _i, C.POINTER(C.c_float)]),
    "fh_rec_get_precision": (_i, [_vp]),
    "fh_det_set_cus": (_i, [_vp, _i]),
    "fh_rec_set_cus": (_i, [_vp, _i]),
    "fh_debug_streamk": (_i, [_i, _i]),
    "fh_debug_wino_slots": (_i, [_i]),
    "fh_set_graph_replay": (_i, [_i]),
    "fh_det_workspace_dev": (_vp, [_vp]),
    "fh_det_graph_stats": (_i, [_vp, _vp]),
    "fh_rec_graph_stats": (_i, [_vp, _vp]),
    "fh_det_set_fused_stem": (_i, [_vp, _i]),
    "fh_rec_set_fused_stem": (_i, [_vp, _i]),
    "fh_det_set_fused_front": (_i, [_vp, _i]),
    "fh_det_set_halo_conv": (_i, [_vp, _i]),
    "fh_resize_u8c3_dev": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "fh_conv_forward_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fh_conv_winograd_dev": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fh_debug_wino2_clock_mhz": (_d, []),
    "fh_conv_wino2_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fh_conv_wino2_ex_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fh_conv_wt_rows": (_i, [_i]),
    "fh_conv_pack_weights": (_i, [_vp, _i, _i, _i, _vp]),
    "fh_conv_kpad": (_i, [_i]),
}

_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = build()
        # FACEHIP_LIB: a diagnostic build of the same library (phase stamps: scripts/wino2_prof.sh) loaded INSTEAD of the in-tree one —
        # a side copy, so that a failed diagnostic run can never leave a non-production library installed
        path = os.environ.get("FACEHIP_LIB") or path
        # PyTorch (device memory / streams / torch.distributed plumbing) bundles its own HIP
        # runtime.  It has to be loaded FIRST so that libfacehip.so binds to the same runtime
        # instance; two runtimes in one process do not share a device context.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def last_error() -> str:
    return lib().fh_last_error().decode(errors="replace")


class FaceHipError(RuntimeError):
    pass


def check(rc: int, what: str) -> int:
    if rc < 0:
        raise FaceHipError(f"{what} failed ({rc}): {last_error()}")
    return rc
