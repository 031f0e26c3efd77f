"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol that
include/facehip.h declares, the C++ ONNX reader + planner (no GPU calls), the reference-shaped
error behaviour of the Python mirror classes, and the multi-rank host logic on gloo (world 2).
"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd import _lib
from facerecognizeonnx_amd.synth import models
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "facehip.h")).read()
    declared = set(re.findall(r"FH_API\s+[\w\s\*]+?\b(fh_\w+)\s*\(", hdr))
    assert len(declared) >= 40
    L = fa.lib()
    for name in declared:
        assert hasattr(L, name), f"libfacehip.so does not export {name}"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert L.fh_version().startswith(b"facehip")
    # sizes the ABI promises
    assert fa.FACE_DTYPE.itemsize == 60 and L.fh_conv_wt_rows(20) == 128 and L.fh_conv_kpad(27) == 32


def test_compare_faces_is_host_code_and_matches_reference_semantics():
    r = fa.FaceRecognizer
    e0 = np.zeros(512, np.float32); e0[0] = 1
    e1 = np.zeros(512, np.float32); e1[1] = 1
    assert r.compareFaces(e0, e0) == 1.0 and r.compareFaces(e0, e1) == 0.5 and r.compareFaces(e0, -e0) == 0.0
    assert r.compareFaces(e0, e1[:10]) == 0.0 and r.compareFaces([], []) == 0.0      # size mismatch / empty -> 0
    v = np.random.default_rng(0).standard_normal(512).astype(np.float32)
    w = np.random.default_rng(1).standard_normal(512).astype(np.float32)
    dot = np.float32(0)
    for a, b in zip(v, w):                                                            # sequential fp32, as the reference
        dot = np.float32(dot + np.float32(a * b))
    assert r.compareFaces(v, w) == np.float32((dot + np.float32(1)) / np.float32(2))


def test_unloaded_and_bad_models_fail_like_the_reference(tmp_path):
    det, rec = fa.FaceDetector(), fa.FaceRecognizer()
    img = util.frames_u8(1, 32, 32)[0]
    assert len(det.detect(img)) == 0 and rec.extractFeature(img, fa.FaceBox()).size == 0   # "Model not loaded!"
    assert rec.extractFeatureSimple(img).size == 0
    assert not det.loadModel(str(tmp_path / "nope.onnx")) and "cannot open" in _lib.last_error()
    junk = tmp_path / "junk.onnx"
    junk.write_bytes(b"\x0a\x03abc\xff\xff\xff")
    assert not rec.loadModel(str(junk)) and _lib.last_error()
    trunc = tmp_path / "trunc.onnx"
    trunc.write_bytes(open(os.path.join(util.GOLDEN, "tiny_scrfd.onnx"), "rb").read()[:5000])
    assert not det.loadModel(str(trunc))


def test_planner_iresnet_fusions(models_dir):
    for fold in (True, False):
        d = fa.plan_describe(util.tiny_iresnet(models_dir, fold_bn=fold), 112, 112)
        ops = [l for l in d.splitlines() if re.match(r"^\d+ ", l)]
        # 1 stem + 5 blocks x (conv1, conv2) + 4 downsample convs + FC; every BN / PRelu / Add is fused
        assert len(ops) == 1 + 10 + 4 + 1, d
        assert sum("CONV" in o for o in ops) == 15 and sum("GEMM" in o for o in ops) == 1
        assert sum("+prelu" in o for o in ops) == 6 and sum("+res" in o for o in ops) == 5
        assert sum("bn2nd" in o for o in ops) == 6 and sum("bn2nd-only" in o for o in ops) == 1
        assert "out 683 [1x512]" in d
    full = fa.plan_describe(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50), 112, 112)
    assert "ops 54" in full
    gmac = float(re.search(r"GMAC/image ([\d.]+)", full).group(1))
    assert abs(gmac - 6.309) < 0.005                                       # 6.309 GMAC / face


def test_planner_scrfd_and_dynamic_input_defaults(models_dir):
    dyn = util.tiny_scrfd(models_dir, hw=None)
    d = fa.plan_describe(dyn, 640, 640)                                    # dynamic H/W -> caller's default (640)
    assert d.startswith("input 640x640") and "out score_8 [12800x1]" in d and "out kps_32 [800x10]" in d
    st = fa.plan_describe(util.tiny_scrfd(models_dir, hw=128), 640, 640)   # static shape wins over the default
    assert st.startswith("input 128x128") and "out bbox_16 [128x4]" in st
    assert st.count("+res(up2x)") == 2 and st.count("[merged x3]") == 3 and "UPSAMPLE" not in st and " ADD " not in st
    full = fa.plan_describe(models.cached("det_500m_seed100.onnx", models.make_det_500m), 640, 640)
    assert abs(float(re.search(r"GMAC/image ([\d.]+)", full).group(1)) - 0.7335) < 0.001
    pre = fa.plan_describe(models.make_predecoded_det(os.path.join(models_dir, "pre.onnx"), 64, 7, True), 640, 640)
    assert "out dets [64x15]" in pre


def test_planner_folds_dynamic_shape_subgraphs(models_dir):
    """Dynamic-axes exports size their Resize with Shape -> Slice -> Concat sub-graphs (what the public det_500m.onnx
    is expected to contain): with the input size fixed at load time the planner folds them to constants."""
    from oracle import onnx_min, oracle
    from oracle import torch_graph as torch_ref
    dyn = models.make_scrfd(os.path.join(models_dir, "s_dynresize.onnx"), (1, 2, 1, 2), (8, 8, 16, 24, 32, 48), 8, 16, seed=2,
                            cls_bias=-2.0, dynamic_resize=True)
    ref = util.tiny_scrfd(models_dir, hw=None)
    g = onnx_min.load(dyn)
    assert {"Shape", "Slice", "Concat", "Cast"} <= {n.op for n in g.nodes}
    a, b = fa.plan_describe(dyn, 96, 160), fa.plan_describe(ref, 96, 160)
    strip = lambda d: [l for l in d.splitlines() if re.match(r"^\d+ |^out ", l)]
    assert strip(a) == strip(b) and a.count("+res(up2x)") == 2            # identical plan, sub-graphs gone
    x = np.random.default_rng(0).uniform(-1, 1, (1, 3, 64, 96)).astype(np.float32)
    oa, ob = oracle.run_graph(g, {"input.1": x}), torch_ref.run_graph(g, {"input.1": x})
    for k in oa:
        np.testing.assert_allclose(oa[k], ob[k], rtol=2e-5, atol=2e-5)


def test_onnx_writer_reader_round_trip(models_dir):
    from oracle import onnx_min
    p = util.tiny_scrfd(models_dir, hw=None)
    g = onnx_min.load(p)
    assert g.inputs[0][1] == [1, 3, -1, -1] and len(g.outputs) == 9          # dim_param -> -1, like ORT reports
    conv = next(n for n in g.nodes if n.op == "Conv")
    assert conv.attrs["kernel_shape"] == [3, 3] and conv.attrs["strides"] == [2, 2] and conv.attrs["group"] == 1
    rs = next(n for n in g.nodes if n.op == "Resize")
    assert rs.attrs["mode"] == "nearest" and list(g.inits[rs.inputs[2]]) == [1, 1, 2, 2]
    w = g.inits[conv.inputs[1]]
    assert w.dtype == np.float32 and w.shape == (8, 3, 3, 3)


def test_shard_ranges_cover_everything():
    from facerecognizeonnx_amd.distributed import shard_range
    for n in (0, 1, 7, 128, 1000003):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from facerecognizeonnx_amd.distributed import shard_range, allgather_queries, allgather_topk
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(0)
G, Q, k, dim = 4000, 6, 5, 32
gal = rng.standard_normal((G, dim)).astype(np.float32); gal /= np.linalg.norm(gal, axis=1, keepdims=True)
gal[3100] = gal[40]                                              # duplicate rows on different shards: index tie-break
allq = gal[[40, 999, 2500, 3999, 7, 3100]].copy()
b, e = shard_range(Q, rank, world)
q = allgather_queries(torch.from_numpy(allq[b:e]))               # each rank contributes Q/world queries
assert np.array_equal(q.numpy(), allq)
gb, ge = shard_range(G, rank, world)                             # row-sharded gallery
sc = (q.numpy() @ gal[gb:ge].T + 1) / 2
order = np.lexsort((np.arange(ge - gb)[None].repeat(Q, 0), -sc), axis=1)[:, :k]
ls = np.take_along_axis(sc, order, 1).astype(np.float32); li = (order + gb).astype(np.int32)
def cpu_merge(ps, pi, k):                                        # test-side checker standing in for the GPU merge kernel (no GPU here)
    s = ps.permute(1, 0, 2).reshape(Q, -1).numpy(); i = pi.permute(1, 0, 2).reshape(Q, -1).numpy()
    o = np.lexsort((i, -s.astype(np.float64)), axis=1)[:, :k]
    return torch.from_numpy(np.take_along_axis(s, o, 1)), torch.from_numpy(np.take_along_axis(i, o, 1))
s, i = allgather_topk(torch.from_numpy(ls), torch.from_numpy(li), k, merge=cpu_merge)
try:                                                             # the product merge is a HIP kernel: on a CPU tensor it must refuse, not fall back
    allgather_topk(torch.from_numpy(ls), torch.from_numpy(li), k)
    raise SystemExit("merge_topk_dev accepted CPU tensors")
except RuntimeError as e:
    assert "HIP kernel" in str(e)
full = (allq @ gal.T + 1) / 2
ref = np.lexsort((np.arange(G)[None].repeat(Q, 0), -full.astype(np.float32)), axis=1)[:, :k]
assert np.array_equal(i.numpy(), ref), (rank, i.numpy(), ref)
assert list(i.numpy()[0][:2]) == [40, 3100]
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_gallery_topk_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_comm_boundary_loads_rccl_and_rejects_bad_arguments_without_a_gpu():
    """The exchange step behind the C ABI (include/facehip.h fh_comm_*): librccl is dlopen'ed on first use, a unique id is 128 bytes
    (ncclUniqueId) and differs per call; argument errors come back as NULL / FH_ERR_ARG with a message — no compute call is made
    here (no GPU in this container)."""
    ids = [fa.Comm.unique_id() for _ in range(2)]
    assert all(len(i) == 128 for i in ids) and ids[0] != ids[1] and any(ids[0])
    L = fa.lib()
    buf = (ctypes.c_ubyte * 128).from_buffer_copy(ids[0])
    for rank, world, dev in ((2, 2, 0), (-1, 1, 0), (0, 0, 0), (0, 1, -1)):
        assert not L.fh_comm_create(rank, world, ctypes.cast(buf, ctypes.c_void_p), dev)
        assert "bad argument" in _lib.last_error()
    assert not L.fh_comm_create(0, 1, None, 0)
    assert L.fh_comm_rank(None) == -1 and L.fh_comm_world(None) == -1          # FH_ERR_ARG
    assert L.fh_gallery_topk_sharded_dev(None, None, None, 1, 1, None, None, None) == -1
    assert L.fh_comm_allgather_f32_dev(None, None, None, 4, None) == -1
    assert L.fh_timing_num_tags() == 13


def test_round5_test_entry_points_validate_their_arguments_without_a_gpu():
    """Argument errors of the kernel test entry points added in round 5 come back as FH_ERR_ARG before anything touches a device."""
    L = fa.lib()
    one = ctypes.c_void_p(16)                                  # a non-null token: every call below must fail on its sizes first
    assert L.fh_conv_wino2_ex_dev(None, None, None, None, None, None, None, None, None, 0, None, None, None, 1, 8, 8, 64, 64, 0, 0, None) == -1
    assert L.fh_conv_wino2_ex_dev(one, one, None, None, None, one, None, None, None, 0, None, None, None, 0, 8, 8, 64, 64, 0, 0, None) == -1   # empty batch
    assert L.fh_conv_wino2_ex_dev(one, one, None, None, None, one, None, None, None, 0, None, None, None, 1, 8, 8, 32, 64, 0, 0, None) == -1   # cin != 64
    assert L.fh_conv_wino2_ex_dev(one, one, None, None, None, one, None, None, None, 0, None, None, None, 1, 8, 8, 64, 48, 0, 0, None) == -1   # cout % 64
    assert L.fh_conv_wino2_ex_dev(one, one, None, None, None, one, one, None, None, 0, None, None, None, 1, 8, 8, 64, 64, 0, 0, None) == -1    # out2 without s2 / t2
    assert L.fh_conv_wino2_ex_dev(one, one, None, None, None, None, None, None, None, 4, one, one, one, 1, 8, 8, 64, 30, 0, 0, None) == -1     # more than 3 merged outputs
    assert L.fh_conv_wino2_ex_dev(one, one, None, None, None, None, None, None, None, 2, None, one, one, 1, 8, 8, 64, 30, 0, 0, None) == -1    # merged outputs without pointers
    assert "fh_conv_wino2_dev" in _lib.last_error()
    assert L.fh_debug_wino_slots(24) == 0 and L.fh_debug_wino_slots(0) == 0


def test_public_header_is_plain_c_and_the_integration_snippet_compiles(tmp_path):
    """include/facehip.h is the drop-in boundary: it must be consumable from C (cgo / JNI / ctypes-style bindings), not only from C++.
    The sharded-gallery loop of INTEGRATION.md section 4, written out as C99, compiles with -Wall -Wextra -Werror -pedantic and links
    against libfacehip.so."""
    import shutil
    if not shutil.which("gcc"):
        pytest.skip("no host C compiler")
    src = tmp_path / "snippet.c"
    src.write_text(r"""
#include <stddef.h>
#include "facehip.h"
int sharded_match(fh_det* det, fh_rec* rec, const unsigned char* d_frames, int n, int rows, int cols, int F, fh_face* d_faces, int* d_frame_of,
                  float* d_emb, const float* shard_rows, long long shard_n, long long first_global_row, int rank, int world, int local_rank,
                  float* d_scores, int* d_idx, void* stream) {
    unsigned char id[FH_COMM_ID_BYTES];
    fh_comm* comm;
    fh_gallery* g;
    int nq, rc;
    if (rank == 0 && fh_comm_unique_id(id) != FH_OK) return -1;
    /* ... the caller broadcasts id to the other ranks here ... */
    comm = fh_comm_create(rank, world, id, local_rank);
    if (!comm) return -2;
    g = fh_gallery_create(512);
    if (fh_gallery_upload(g, shard_rows, shard_n, 0, first_global_row) < 0) return -3;
    nq = fh_pipeline_run_dev(det, rec, d_frames, n, rows, cols, cols * 3, (long long)rows * cols * 3, .5f, .4f, F, d_faces, d_frame_of, d_emb, stream);
    if (nq < 0) return -4;
    rc = fh_gallery_topk_sharded_dev(g, comm, d_emb, n * F, 16, d_scores, d_idx, stream);
    if (rc != fh_comm_world(comm) * n * F) rc = -5;
    fh_gallery_destroy(g);
    fh_comm_destroy(comm);
    return rc;
}
int main(void) { return fh_version() == NULL; }
""")
    exe = tmp_path / "snippet"
    lib_dir = os.path.join(ROOT, "facerecognizeonnx_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", lib_dir, "-lfacehip", "-Wl,-rpath," + lib_dir]
    fa.lib()                                                   # (builds libfacehip.so if it is missing)
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert b.returncode == 0, b.stdout[-3000:]
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 0, r.stdout[-2000:]               # fh_version() works without a GPU


def _affine_graph(path, H=24, W=20, C=12, Cout=8):
    """conv -> Mul(scalar) -> Add(per-channel) -> Relu -> Dropout -> conv -> Identity -> Sub(scalar) -> Div(per-channel): the
    element-wise constant ops and pass-through nodes exporters leave in graphs (SCRFD's Scale layers, normalisation nodes)."""
    from facerecognizeonnx_amd.synth.onnx_writer import OnnxBuilder
    rng = np.random.default_rng(5)
    b = OnnxBuilder("affine")
    x = b.add_input("input", [1, 3, H, W])
    def conv(x, cout, cin, k):
        return b.node("Conv", [x, b.init(b.uid("w"), (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)),
                               b.init(b.uid("b"), (rng.standard_normal(cout) / 10).astype(np.float32))],
                      kernel_shape=[k, k], pads=[k // 2] * 4, strides=[1, 1])
    y = conv(x, C, 3, 3)
    y = b.node("Mul", [y, b.init("scale0", np.array(1.7, np.float32))])
    y = b.node("Add", [b.init("shift0", rng.standard_normal((1, C, 1, 1)).astype(np.float32)), y])       # constant first
    y = b.node("Relu", [y])
    y = b.node("Dropout", [y], ratio=0.4)                                                               # inference: pass-through
    y = conv(y, Cout, C, 1)
    y = b.node("Identity", [y])
    y = b.node("Sub", [y, b.init("mean1", np.array([0.25], np.float32))])
    y = b.node("Div", [y, b.init("std1", rng.uniform(0.5, 2.0, (Cout, 1, 1)).astype(np.float32))])
    y = b.node("Transpose", [y], perm=[0, 2, 3, 1])
    b.node("Reshape", [y, b.init("shape", np.array([-1, Cout], np.int64))], outputs=["out"])
    b.add_output("out", ["A", Cout])
    return b.save(path)


def test_planner_folds_constant_mul_add_sub_div(tmp_path):
    import facerecognizeonnx_amd as fa
    from oracle import oracle
    from oracle import torch_graph as torch_ref
    path = _affine_graph(str(tmp_path / "affine.onnx"))
    desc = fa.plan_describe(path, 24, 20)
    assert desc.splitlines()[0].split("ops ")[1].startswith("2 ")          # everything folded into the two convolutions
    rng = np.random.default_rng(6)
    x = rng.standard_normal((1, 3, 24, 20)).astype(np.float32)
    od = oracle.OracleDetector(); assert od.loadModel(path)
    o = od.run_network(x[0])[0]
    t = np.asarray(torch_ref.run_graph(path, {"input": x})["out"])
    np.testing.assert_allclose(o, t.reshape(o.shape), rtol=1e-5, atol=1e-5)  # the oracle's literal evaluation vs fp64


def test_mobilefacenet_plans_and_oracle_matches_fp64(models_dir):
    """The buffalo_s / buffalo_sc recogniser (w600k_mbf) loads through the same planner; the oracle evaluates its graph."""
    import facerecognizeonnx_amd as fa
    from facerecognizeonnx_amd.synth import models
    from oracle import oracle
    from oracle import torch_graph as torch_ref
    from tests import util
    full = fa.plan_describe(models.cached("w600k_mbf_seed300.onnx", models.make_w600k_mbf), 112, 112)
    head = full.splitlines()[0]
    assert "ops 50" in head and 0.40 < float(head.split("GMAC/image ")[1].split()[0]) < 0.47      # MobileFaceNet: ~0.44 GMAC
    assert "GCONV k3s1 56x56x128 -> 56x56x128+prelu" in full and "DWGLOBAL k7s1 7x7x512 -> 1x1x512" in full
    path = util.tiny_mbf(models_dir, fold_bn=False)
    x = np.random.default_rng(3).standard_normal((1, 3, 112, 112)).astype(np.float32)
    g = oracle.OracleRecognizer(); assert g.loadModel(path)
    o = oracle.run_graph(g.g, {g.g.inputs[0][0]: x})[g.g.outputs[0][0]]
    t = np.asarray(torch_ref.run_graph(path, {g.g.inputs[0][0]: x})[g.g.outputs[0][0]])
    np.testing.assert_allclose(o, t, rtol=1e-4, atol=1e-4)


def test_planner_keeps_plain_output_alive_for_the_winograd_bn_link(tmp_path):
    """A Winograd conv1 may read its block input's PLAIN tensor (and apply bn1 itself) instead of the BatchNorm'ed second
    output (POp::bn_src).  When the block's shortcut conv is written in front of conv1, the plain tensor's last listed reader
    comes before conv1: the planner must still keep it alive — and un-aliased — up to conv1."""
    path = models.make_iresnet(str(tmp_path / "ds_first.onnx"), (1, 1, 1, 1), (32, 128, 128, 128), 112, 64, seed=5, downsample_first=True)
    desc = fa.plan_describe(path, 112, 112)
    ops = [l for l in desc.splitlines() if re.match(r"^\d+ ", l)]
    tens = {int(m.group(1)): (int(m.group(2)) * int(m.group(3)) * int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)))
            for m in re.finditer(r"tensor t(\d+) (\d+)x(\d+)x(\d+) off (\d+) live (-?\d+)\.\.(-?\d+)", desc)}
    links = 0
    for l in ops:
        m = re.search(r"bn<-op(\d+)", l)
        if not m:
            continue
        i, prod = int(l.split()[0]), int(m.group(1))
        plain = int(re.search(r"out t(-?\d+)", ops[prod]).group(1))
        shortcut_between = any(re.search(rf"\[in t{plain} ", ops[j]) for j in range(prod + 1, i))
        elems, off, first, last = tens[plain]
        assert first <= prod and last >= i, (l, tens[plain])
        for t, (e2, o2, f2, l2) in tens.items():                      # nothing else may occupy its floats while it is live
            if t != plain and not (l2 < first or last < f2):
                assert o2 + e2 <= off or off + elems <= o2, (plain, t)
        links += shortcut_between
    assert links >= 3                                                  # the stage-opening blocks: shortcut conv sits between


@pytest.mark.parametrize("ds_first", [False, True])
def test_planner_keeps_the_shortcut_input_alive_for_the_folded_form(tmp_path, ds_first):
    """POp::sc_src: a strided block's 3x3 convolution may run the 1x1 shortcut inside its own K loop, i.e. read the SHORTCUT's input
    at its own position — which the op list does not show when the shortcut comes first.  That tensor must stay live and un-aliased."""
    path = models.make_iresnet(str(tmp_path / f"sc_{int(ds_first)}.onnx"), (1, 2, 1, 1), (32, 64, 128, 128), 112, 64, seed=9, downsample_first=ds_first)
    desc = fa.plan_describe(path, 112, 112)
    ops = [l for l in desc.splitlines() if re.match(r"^\d+ ", l)]
    tens = {int(m.group(1)): (int(m.group(2)) * int(m.group(3)) * int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)))
            for m in re.finditer(r"tensor t(\d+) (\d+)x(\d+)x(\d+) off (\d+) live (-?\d+)\.\.(-?\d+)", desc)}
    links = 0
    for l in ops:
        m = re.search(r"sc<-op(\d+)", l)
        if not m:
            continue
        i, sc = int(l.split()[0]), int(m.group(1))
        assert " k3s2 " in l and "+res" in l and " k1s2 " in ops[sc]
        src = int(re.search(r"\[in t(-?\d+)", ops[sc]).group(1))
        elems, off, first, last = tens[src]
        assert last >= i, (l, tens[src])
        for t, (e2, o2, f2, l2) in tens.items():
            if t != src and not (l2 < first or last < f2):
                assert o2 + e2 <= off or off + elems <= o2, (src, t)
        links += 1
    assert links == 4                                                  # one per stage


def test_wino2_layout_model_is_conflict_free_and_exact():
    """CPU model of conv_wino2.hip's data movement (scripts/wino2_banks.py mirrors the kernel's index arithmetic one to one): every
    ds_read_b128 of the K loop touches 16 different bank quads per service group, and the emulated kernel — swizzled parity planes,
    lane -> tile map, packed weight order, F(2x2,3x3) transforms — equals a direct convolution (the Conv nodes of
    face_recognizer.cpp:279-283's graph) to rounding, incl. ragged tile groups and the merged-head channel count."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wino2_banks", os.path.join(ROOT, "scripts", "wino2_banks.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.bank_report() == (1, 1.0)
    assert mod.emulate(1, 10, 13, 64) < 1e-12
    assert mod.emulate(1, 8, 8, 30) < 1e-12
