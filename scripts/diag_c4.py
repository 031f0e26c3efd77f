"""Diagnostic for the C4 composition test: where do the GPU's record and the oracle's record of the best face differ, and does
the aligned crop change?"""
import numpy as np, torch
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
from oracle import oracle
from tests import util
fa.lib().fh_init(0); oracle.set_threads(16)
dpath = models.cached("det_500m_seed100.onnx", models.make_det_500m)
rpath = models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)
det = fa.FaceDetector(); rec = fa.FaceRecognizer(); odet = oracle.OracleDetector(); orec = oracle.OracleRecognizer()
assert det.loadModel(dpath) and rec.loadModel(rpath) and odet.loadModel(dpath) and orec.loadModel(rpath)
B = 64
frames = np.concatenate([util.frames_u8(B // 2, 640, 640, seed=401), util.frames_u8(B // 2, 640, 640, seed=402, smooth=True)])
fd = torch.from_numpy(frames).cuda()
faces = torch.zeros((B, 15), device="cuda"); fo = torch.full((B,), -1, dtype=torch.int32, device="cuda"); emb = torch.zeros((B, 512), device="cuda")
total = fa.pipeline_run_dev(det, rec, fd.data_ptr(), B, 640, 640, 1, faces.data_ptr(), fo.data_ptr(), emb.data_ptr(), 0.5, 0.4)
torch.cuda.synchronize()
recs = faces.cpu().numpy().view(np.uint8).reshape(B, 60).copy().view(fa.FACE_DTYPE).reshape(B)[:total]
frame_of = fo.cpu().numpy()[:total]; e = emb.cpu().numpy()[:total]
print("total", total)
for i in range(0, total, 3):
    b = int(frame_of[i]); ref = odet.detect(frames[b], 0.5, 0.4)
    r = ref[0]; g = recs[i]
    ca = oracle.align_face(frames[b], g); cb = oracle.align_face(frames[b], r)
    nd = int((ca != cb).sum()) if ca is not None and cb is not None else -1
    comp = orec.extractFeature(frames[b], r); own = orec.extractFeature(frames[b], g)
    print(b, "n_ref", len(ref), "score", g["score"], r["score"], "box", [int(g[k]) - int(r[k]) for k in "xywh"], "lm maxdiff", float(np.abs(g["lm"] - r["lm"]).max()),
          "crop bytes differing", nd, "1-cos own", 1 - float(e[i] @ own), "comp", 1 - float(e[i] @ comp), "lm", g["lm"][:4])
