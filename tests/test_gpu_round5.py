"""GPU parity tests added in round 5: the HEADLINE in the form bench.py times it (128 frames, two HIP streams, three batches in flight)
against the oracle's own composition; the sharded gallery exchange behind the C ABI (fh_comm_* over RCCL); the channel-sliced fused
Winograd transform of the 28x28 stage; small-batch GEMM tiles.

Same bars as tests/test_gpu_parity.py: integer / byte / index work bit-exact, fp32 network outputs within the tolerance written
beside each assert, embeddings within 1e-3 cosine of the oracle (north star).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import facerecognizeonnx_amd as fa            # noqa: E402
from facerecognizeonnx_amd.synth import models  # noqa: E402
from oracle import oracle                     # noqa: E402
from tests import util                        # noqa: E402
from tests.test_gpu_parity import dev         # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real device: the product path has no CPU fallback")
    fa.lib().fh_init(0)
    oracle.set_threads(min(16, os.cpu_count() or 8))


def _recs(faces_t, total):
    return faces_t.cpu().numpy().view(np.uint8).reshape(-1, 60).copy().view(fa.FACE_DTYPE).reshape(-1)[:total]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("F", [1, 4])
def test_headline_streaming_form_matches_oracle_composition(F):
    """BASELINE.json's metric in the form bench.py times it (bench.py `step()` of the pipelined branch): batches of 128 frames of
    640 x 640, full-size det_500m + w600k_r50, `fh_pipeline_submit_dev` with the detector on one HIP stream and the recogniser on
    another, a ring of 3 result slots with event back-pressure (3 batches in flight), thresholds 0.5 / 0.4 (face_detector.h:20),
    the first F faces of every frame embedded.  The caller being restated is main.cpp:88-114 / 221-238 (detect, for each face
    extractFeature) over a batch.

    Checked against the ORACLE's own composition — `odet.detect` (face_detector.cpp:139-222) -> its first F faces (NMS output is
    score-descending, face_detector.cpp:356-384) -> `orec.extractFeature` (face_recognizer.cpp:236-304) — on 8 frames spread over the
    batch (both ends, the middle, the random and the smooth half): records +-1 px / 1e-4 / 1e-2 px, embeddings 1 - cos < 1e-3 (north
    star).  Every other slot of every batch in flight is compared BITWISE with the serial entry point `fh_pipeline_run_dev`."""
    dpath = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    rpath = models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)
    det = fa.FaceDetector(); rec = fa.FaceRecognizer(); odet = oracle.OracleDetector(); orec = oracle.OracleRecognizer()
    assert det.loadModel(dpath) and rec.loadModel(rpath) and odet.loadModel(dpath) and orec.loadModel(rpath)
    B, RING, NB = 128, 3, 4
    host = [np.concatenate([util.frames_u8(B // 2, 640, 640, seed=900 + 2 * k), util.frames_u8(B // 2, 640, 640, seed=901 + 2 * k, smooth=True)])
            for k in range(2)]
    batches = [dev(h) for h in host]
    # serial form first (one stream, host sync per batch): the bitwise reference for every slot
    serial = []
    for k in range(2):
        f = torch.zeros((B * F, 15), device="cuda"); o = torch.full((B * F,), -1, dtype=torch.int32, device="cuda"); e = torch.zeros((B * F, 512), device="cuda")
        t = fa.pipeline_run_dev(det, rec, batches[k].data_ptr(), B, 640, 640, F, f.data_ptr(), o.data_ptr(), e.data_ptr(), 0.5, 0.4)
        torch.cuda.synchronize()
        serial.append((t, f[:t].clone(), o[:t].clone(), e[:t].clone()))
    # streaming form exactly as bench.py drives it
    s_det, s_rec = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)
    rf = [torch.zeros((B * F, 15), device="cuda") for _ in range(RING)]
    ro = [torch.full((B * F,), -1, dtype=torch.int32, device="cuda") for _ in range(RING)]
    re_ = [torch.zeros((B * F, 512), device="cuda") for _ in range(RING)]
    rt = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(RING)]
    done = [None] * RING
    kept = []
    for k in range(NB):
        slot = k % RING
        if done[slot] is not None:
            s_det.wait_event(done[slot])
            done[slot].synchronize()                       # (the test copies the slot's results out before it is reused)
            kept[k - RING] = tuple(x.clone() for x in kept[k - RING])
        n = fa.pipeline_submit_dev(det, rec, batches[k % 2].data_ptr(), B, 640, 640, F, rf[slot].data_ptr(), ro[slot].data_ptr(),
                                   re_[slot].data_ptr(), rt[slot].data_ptr(), s_det.cuda_stream, s_rec.cuda_stream, 0.5, 0.4)
        done[slot] = torch.cuda.Event(); done[slot].record(s_rec)
        kept.append((rf[slot], ro[slot], re_[slot], rt[slot]))
        assert n == serial[k % 2][0], (k, n, serial[k % 2][0])
    torch.cuda.synchronize()
    for k in range(NB):
        t, sf, so, se = serial[k % 2]
        f, o, e, tt = kept[k]
        assert int(tt.item()) == t
        assert torch.equal(f[:t].view(torch.int32), sf.view(torch.int32)), k
        assert torch.equal(o[:t], so) and torch.equal(e[:t], se), k
    # oracle composition on 8 frames of the LAST batch in flight
    kb = (NB - 1) % 2
    t, sf, so, se = serial[kb]
    f, o, e, _ = kept[NB - 1]
    recs = _recs(f, t); frame_of = o.cpu().numpy()[:t]; emb = e.cpu().numpy()[:t]
    assert t >= B // 2                                                    # most synthetic frames fire above 0.5
    assert np.all(np.diff(frame_of) >= 0) and np.bincount(frame_of, minlength=B).max() <= F
    assert np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    worst = 0.0; checked = 0
    for b in (0, 17, 45, 63, 64, 90, 111, 127):
        ref = odet.detect(host[kb][b], 0.5, 0.4)
        mine = np.where(frame_of == b)[0]
        want = min(F, len(ref))
        # the pipeline embeds the first F post-NMS faces; a face whose score is within 1e-4 of the threshold may flip
        assert abs(len(mine) - want) <= (1 if any(abs(float(r["score"]) - 0.5) < 1e-4 for r in ref[:F + 1]) else 0), (b, len(mine), want)
        for j, i in enumerate(mine[:want]):
            miss, _ = util.match_records(recs[i:i + 1], ref[j:j + 1])
            assert not miss, (b, j, recs[i], ref[j])
            comp = orec.extractFeature(host[kb][b], ref[j])              # oracle align + embed on the ORACLE's record
            assert comp.size == 512
            worst = max(worst, 1.0 - float(np.dot(emb[i].astype(np.float64), comp.astype(np.float64))))
            checked += 1
    assert checked >= 6, checked
    assert worst < 1e-3, worst                                           # north-star bar


_COMM_WORKER = r"""
import os, sys, ctypes
sys.path.insert(0, sys.argv[1])
import numpy as np
import facerecognizeonnx_amd as fa
# the communicator is the FIRST GPU call of this process (fh_comm_create binds the device); no torch.distributed anywhere:
# the exchange runs behind the C ABI on librccl
uid = fa.Comm.unique_id()
comm = fa.Comm(0, 1, uid, 0)
assert fa.lib().fh_comm_rank(comm.handle) == 0 and fa.lib().fh_comm_world(comm.handle) == 1
import torch                                              # device buffers only
torch.cuda.set_device(0)
from oracle import oracle
fa._lib.check(fa.lib().fh_init(0), "fh_init")
oracle.set_threads(min(16, os.cpu_count() or 8))
assert "torch.distributed" not in sys.modules or not torch.distributed.is_initialized()
side = torch.cuda.Stream()
# ---- (a) small shard, odd sizes, non-default stream: Q = 7 local queries, k = 5, index base 1000
rng = np.random.default_rng(11)
G, Q, k, base = 6001, 7, 5, 1000
gal = rng.standard_normal((G, 512)).astype(np.float32); gal /= np.linalg.norm(gal, axis=1, keepdims=True)
q = gal[[17, 4100, 3, 6000, 2999, 512, 77]].copy(); gal[4100] = gal[17]
g = fa.Gallery(512); gd = torch.from_numpy(gal).cuda(); g.upload(gd.data_ptr(), G, True, base)
qd = torch.from_numpy(q).cuda()
s = torch.zeros((Q, k), device="cuda"); i = torch.zeros((Q, k), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
assert comm.gallery_topk_sharded_dev(g, qd.data_ptr(), Q, k, s.data_ptr(), i.data_ptr(), side.cuda_stream) == Q
side.synchronize()
rs, ri = oracle.gallery_topk(q, gal, k)
np.testing.assert_allclose(s.cpu().numpy(), rs, atol=2e-6)
assert np.array_equal(i.cpu().numpy().astype(np.int64) - base, ri), (i.cpu().numpy(), ri)
assert i[0, :2].tolist() == [base + 17, base + 4100]
# the single-gallery call gives the same bits (same scan, same merge kernel)
s1 = torch.zeros((Q, k), device="cuda"); i1 = torch.zeros((Q, k), dtype=torch.int32, device="cuda")
g.topk_dev(qd.data_ptr(), Q, k, s1.data_ptr(), i1.data_ptr()); torch.cuda.synchronize()
assert torch.equal(s, s1) and torch.equal(i, i1)
# plain all-gather entry point
out = torch.zeros_like(qd)
assert comm.allgather_f32_dev(qd.data_ptr(), out.data_ptr(), qd.numel(), side.cuda_stream) == 1
side.synchronize(); assert torch.equal(out, qd)
# more than 256 gathered queries: the scan is chunked
Q2 = 300
q2 = rng.standard_normal((Q2, 512)).astype(np.float32); q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
q2d = torch.from_numpy(q2).cuda(); s2 = torch.zeros((Q2, k), device="cuda"); i2 = torch.zeros((Q2, k), dtype=torch.int32, device="cuda")
assert comm.gallery_topk_sharded_dev(g, q2d.data_ptr(), Q2, k, s2.data_ptr(), i2.data_ptr()) == Q2
torch.cuda.synchronize()
rs2, ri2 = oracle.gallery_topk(q2, gal, k)
np.testing.assert_allclose(s2.cpu().numpy(), rs2, atol=2e-6)
assert (i2.cpu().numpy().astype(np.int64) - base != ri2).mean() < 0.01
del g, gd
# ---- (b) config C5's per-rank workload: the LAST of 8 shards of a 10 M-row gallery (1.25 M rows, index base 8.75 M), 64 queries, k = 16
TOTAL, WORLD, RANK8 = 10_000_000, 8, 7
from facerecognizeonnx_amd.distributed import gallery_shard_base
b, e = gallery_shard_base(TOTAL, RANK8, WORLD)
G, Q, k = e - b, 64, 16
rng = np.random.default_rng(5)
gal = np.empty((G, 512), np.float32)
for s0 in range(0, G, 125_000):
    blk = rng.standard_normal((min(125_000, G - s0), 512), dtype=np.float32)
    blk /= np.linalg.norm(blk, axis=1, keepdims=True)
    gal[s0:s0 + len(blk)] = blk
q = rng.standard_normal((Q, 512)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
S = 1 << 20
gal[3] = q[0]; gal[S - 1] = q[0]; gal[S] = q[0]; gal[G - 1] = q[0]
gal[S - 2] = q[1]; gal[S + 1] = q[1]
g = fa.Gallery(512); gd = torch.from_numpy(gal).cuda(); g.upload(gd.data_ptr(), G, True, b); del gd
qd = torch.from_numpy(q).cuda()
s = torch.zeros((Q, k), device="cuda"); i = torch.zeros((Q, k), dtype=torch.int32, device="cuda")
assert comm.gallery_topk_sharded_dev(g, qd.data_ptr(), Q, k, s.data_ptr(), i.data_ptr(), side.cuda_stream) == Q
side.synchronize()
gs, gi = s.cpu().numpy(), i.cpu().numpy().astype(np.int64)
assert gi.min() >= b and gi.max() < e
rs, ri = oracle.gallery_topk(q, gal, k)
np.testing.assert_allclose(gs, rs, atol=2e-6)
li_ = gi - b
exact = lambda idx: (np.einsum("qkd,qd->qk", gal[idx].astype(np.float64), q.astype(np.float64)) + 1.0) / 2.0
diff = li_ != ri
assert np.abs(exact(li_) - exact(ri))[diff].max(initial=0.0) < 1e-6 and diff.mean() < 0.01
assert gi[0, :4].tolist() == [b + 3, b + S - 1, b + S, b + G - 1]
assert gi[1, :2].tolist() == [b + S - 2, b + S + 1]
maps = open("/proc/self/maps").read()
assert "librccl" in maps
comm.close()
print("comm ok; rank swaps", int(diff.sum()))
"""


@pytest.mark.timeout(600)
def test_sharded_gallery_exchange_behind_the_c_abi_world1(tmp_path):
    """SURVEY.md 8(e) behind the drop-in boundary: `fh_comm_create` (librccl, no torch.distributed) as the first GPU call of a fresh
    child, then `fh_gallery_topk_sharded_dev` = queries all-gather -> local scan -> ONE top-k all-gather -> topk_merge_kernel on the
    caller's stream, against `oracle.gallery_topk` ((dot + 1) / 2, face_recognizer.cpp:320-334; score desc, index asc): a small shard
    with a non-zero index base on a non-default stream, 300 queries (chunked scan), and config C5's per-rank shard (1.25 M rows, base
    8.75 M, duplicates across the 2^20-row boundary).  One GPU per box: world = 1 (8 ranks: unmeasured)."""
    script = tmp_path / "comm_worker.py"
    script.write_text(_COMM_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=560)
    assert p.returncode == 0 and "comm ok" in p.stdout, p.stdout[-3000:]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("size", [112, 104])
def test_channel_sliced_fused_transform_on_28x28_maps_matches_oracle_and_unfused(tmp_path, size):
    """Between two Winograd layers on a 28 x 28 map (49 tiles) the output transform of the first and the input transform of the second
    run as ONE kernel per (image, 32-channel slice) with the whole slice in LDS (winograd.hip `wino_slice_kernel`) — also at junctions
    that read a residual and write out1 / out2.  An IResNet whose stage 2 is 28 x 28 x 128 (three blocks: conv -> PReLU -> conv junctions
    without memory traffic, conv -> +residual -> BN -> conv junctions with it), batch 24 (>= 256 tiles per GEMM: Winograd form):
    raw outputs against the oracle's direct fp32 evaluation (the stand-in for ORT's Run, face_recognizer.cpp:279-283) and to 1e-5 of
    scale against the same handle with the fusion switched off (`fh_rec_set_wino_fusion(0)`: separate transform kernels — the same
    arithmetic, but the compiler contracts multiply-adds per kernel, so not bit for bit).  size = 104 makes the stage 26 x 26: still
    7 x 7 tiles, the last tile row / column with two of its four pixels outside the map (stores masked, patch reads zero-filled)."""
    m = size // 4
    path = models.make_iresnet(str(tmp_path / f"s{m}.onnx"), (1, 3, 1, 1), (32, 128, 128, 128), size, 64, seed=21)
    desc = fa.plan_describe(path, size, size)
    assert desc.count(f"k3s1 {m}x{m}x128 -> {m}x{m}x128") >= 5, desc
    rec = fa.FaceRecognizer(); orec = oracle.OracleRecognizer()
    assert rec.loadModel(path) and orec.loadModel(path)
    n = 24
    crops = util.frames_u8(n, size, size, seed=77)
    outs = []
    for fuse in (1, 0):
        assert fa.lib().fh_rec_set_wino_fusion(rec.handle, fuse) == 0
        out = torch.zeros((n, 64), device="cuda"); raw = torch.zeros((n, 64), device="cuda")
        assert rec.embed_aligned_dev(dev(crops).data_ptr(), n, out.data_ptr(), raw.data_ptr()) == n
        torch.cuda.synchronize()
        outs.append(raw.cpu().numpy())
    assert np.abs(outs[0] - outs[1]).max() <= 1e-5 * np.abs(outs[1]).max(), np.abs(outs[0] - outs[1]).max()
    for i in (0, n // 2, n - 1):
        r = oracle.run_graph(orec.g, {orec.g.inputs[0][0]: oracle.rec_preprocess(crops[i])[None]})[orec.g.outputs[0][0]].reshape(-1)
        np.testing.assert_allclose(outs[0][i], r, rtol=2e-4, atol=2e-4 * np.abs(r).max(), err_msg=f"slot {i}")


WINO2_MERGED_CASES = [
    # B, H,  W,  splits of the <= 32 merged channels, activations per group (0 none / 1 ReLU / 3 sigmoid)
    (3, 80, 80, (2, 8, 20), (3, 0, 0)),         # SCRFD's merged heads: score (sigmoid) / bbox / kps of one stride, 30 of 32 channels live
    (2, 20, 20, (3, 8, 19), (3, 0, 1)),         # odd group sizes: channel pairs that straddle two destinations, unaligned 8-byte pairs
    (5, 40, 40, (1, 4, 10), (3, 0, 0)),         # 15 channels: the second 16-channel block is all padding
    (2, 13, 9, (32,), (1,)),                    # one destination, ragged map (tile groups hanging over both borders)
    (1, 20, 20, (5, 27), (0, 3)),               # two destinations
]


@pytest.mark.parametrize("B,H,W,splits,acts", WINO2_MERGED_CASES)
def test_wino2_merged_sibling_epilogue_matches_oracle(B, H, W, splits, acts):
    """The CB = 2 form of conv_wino2.hip in isolation (round-4 advisor finding: it only ever ran inside whole-network tests): sibling
    3x3 convolutions of one 64-channel map evaluated as ONE convolution of <= 32 channels whose channel ranges go to separate
    destinations through separate activations — SCRFD's score / bbox / kps heads per stride (the Conv + Sigmoid nodes at the end of
    session_->Run, face_detector.cpp:179-183).  Each destination against the oracle's direct fp32 convolution of its own filter slice,
    5e-5 absolute as for the plain form; sigmoid outputs 1e-6."""
    cout = sum(splits)
    rng = np.random.default_rng(B * 100 + H + cout)
    x = rng.standard_normal((B, 64, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, 64, 3, 3)) / np.sqrt(64 * 9)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    ref = oracle.conv2d(x, w, None, 1, 1, 1) + bias[None, :, None, None]
    oc0 = np.concatenate([[0], np.cumsum(splits)]).astype(np.int32)
    outs = [torch.full((B, H, W, c), float("nan"), device="cuda") for c in splits]
    ptrs = (ctypes.c_void_p * len(splits))(*[o.data_ptr() for o in outs])
    oact = np.array(acts, np.int32)
    xd, bd = dev(x.transpose(0, 2, 3, 1)), dev(bias)
    w_ohwi = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(cout, 9, 64))
    rc = fa.lib().fh_conv_wino2_ex_dev(xd.data_ptr(), w_ohwi.ctypes.data, bd.data_ptr(), None, None, None, None, None, None, len(splits),
                                       ctypes.cast(ptrs, ctypes.c_void_p), oc0.ctypes.data, oact.ctypes.data, B, H, W, 64, cout, 0, 0, None)
    assert rc == 0, fa._lib.last_error()
    torch.cuda.synchronize()
    for g, (c, a) in enumerate(zip(splits, acts)):
        r = ref[:, oc0[g]:oc0[g + 1]]
        r = np.maximum(r, 0) if a == 1 else 1.0 / (1.0 + np.exp(-r.astype(np.float64))) if a == 3 else r
        got = outs[g].cpu().numpy().transpose(0, 3, 1, 2)
        assert not np.isnan(got).any(), g
        np.testing.assert_allclose(got, r.astype(np.float32), rtol=0, atol=5e-5 if a != 3 else 2e-5, err_msg=f"destination {g}")


def test_wino2_second_output_and_argument_checks():
    """`out2 = out1 * s2 + t2` of the plain form (a following block's BatchNorm written beside the plain output; IResNet's
    `+bn2nd` layers) with and without out1, and the argument checks of the test entry point."""
    B, H, W, C = 2, 24, 24, 64
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32); s2 = rng.uniform(0.5, 1.5, C).astype(np.float32); t2 = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((B, H, W, C)).astype(np.float32)
    ref = oracle.conv2d(x, w, None, 1, 1, 1) + bias[None, :, None, None] + res.transpose(0, 3, 1, 2)
    ref2 = ref * s2[None, :, None, None] + t2[None, :, None, None]
    xd, bd, rd, s2d, t2d = dev(x.transpose(0, 2, 3, 1)), dev(bias), dev(res), dev(s2), dev(t2)
    w_ohwi = np.ascontiguousarray(w.transpose(0, 2, 3, 1).reshape(C, 9, C))
    L = fa.lib()
    for with_out1 in (True, False):
        o1 = torch.full((B, H, W, C), float("nan"), device="cuda"); o2 = torch.full((B, H, W, C), float("nan"), device="cuda")
        rc = L.fh_conv_wino2_ex_dev(xd.data_ptr(), w_ohwi.ctypes.data, bd.data_ptr(), None, rd.data_ptr(), o1.data_ptr() if with_out1 else None,
                                    o2.data_ptr(), s2d.data_ptr(), t2d.data_ptr(), 0, None, None, None, B, H, W, C, C, 0, 0, None)
        assert rc == 0, fa._lib.last_error()
        torch.cuda.synchronize()
        if with_out1:
            np.testing.assert_allclose(o1.cpu().numpy().transpose(0, 3, 1, 2), ref, rtol=0, atol=5e-5)
        else:
            assert torch.isnan(o1).all()          # untouched
        np.testing.assert_allclose(o2.cpu().numpy().transpose(0, 3, 1, 2), ref2, rtol=0, atol=1e-4)
    o1 = torch.zeros((B, H, W, C), device="cuda")
    assert L.fh_conv_wino2_ex_dev(xd.data_ptr(), w_ohwi.ctypes.data, bd.data_ptr(), None, None, o1.data_ptr(), o1.data_ptr(), None, None, 0, None,
                                  None, None, B, H, W, C, C, 0, 0, None) == -1           # second output without scale / shift
    assert L.fh_conv_wino2_ex_dev(xd.data_ptr(), w_ohwi.ctypes.data, bd.data_ptr(), None, None, o1.data_ptr(), None, None, None, 0, None,
                                  None, None, 0, H, W, C, C, 0, 0, None) == -1           # empty batch
    assert L.fh_conv_wino2_ex_dev(xd.data_ptr(), w_ohwi.ctypes.data, bd.data_ptr(), None, None, o1.data_ptr(), None, None, None, 0, None,
                                  None, None, B, H, W, C, 48, 0, 0, None) == -1          # plain layers: whole 64-channel column tiles


@pytest.mark.parametrize("B,H,W,Cin,Cout,slots", [
    (2, 14, 14, 256, 256, 8),          # 128x128 tiles (Cout % 128 == 0 picks them when tiles spill past the slots): 9 tiles per workgroup
    (1, 28, 28, 128, 128, 8),          # 128x64 tiles, K = 128: four chunks per tile, an even number (the tile buffers swap roles per tile)
    (3, 7, 7, 512, 128, 16),           # K = 512
    (2, 13, 10, 160, 64, 8),           # K = 160: FIVE chunks, an odd number — the next tile's first chunk lands in the other buffer
    (64, 14, 14, 128, 64, 8),          # mixed F(4x4) / F(2x2) layout: plane classes change inside a workgroup's walk
    (2, 14, 14, 256, 256, 24),         # three tiles per workgroup, the last walk ragged
])
def test_multi_tile_winograd_gemm_walks_match_oracle_and_single_tile_form(B, H, W, Cin, Cout, slots):
    """`wino_gemm_pers_kernel` (several GEMM tiles per workgroup, the next tile's first chunk requested in front of the store tail,
    counted `vmcnt` wait): with the slot count forced down (fh_debug_wino_slots) every workgroup walks many tiles.  The layer must match
    the oracle's direct convolution at the Winograd bar (2e-4 abs: the Conv nodes inside session_->Run, face_recognizer.cpp:279-283)
    and equal the one-tile-per-workgroup launch BIT FOR BIT (same tile arithmetic in the same order)."""
    rng = np.random.default_rng(B * 1000 + H * 10 + Cin + slots)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = oracle.conv2d(x, w, b, 1, 1, 1)
    ohwi = np.ascontiguousarray(w.transpose(0, 2, 3, 1))
    xd, bd = dev(x.transpose(0, 2, 3, 1)), dev(b)
    L = fa.lib()
    outs = []
    try:
        for s in (slots, 1 << 20):                                         # walk / one tile per workgroup (more slots than tiles)
            assert L.fh_debug_wino_slots(s) == 0
            out = torch.full((B, H, W, Cout), float("nan"), device="cuda")
            assert L.fh_conv_winograd_dev(xd.data_ptr(), ohwi.ctypes.data, bd.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, 0) == 0, fa._lib.last_error()
            outs.append(out.cpu().numpy())
    finally:
        L.fh_debug_wino_slots(0)
    assert np.array_equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0].transpose(0, 3, 1, 2), ref, rtol=0, atol=2e-4)
