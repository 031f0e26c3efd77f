"""Detector and recogniser side by side on disjoint CU sets (hipExtStreamCreateWithCUMask): python scripts/cumask_probe.py [det_cus ...]"""
import ctypes as C, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
L = fa.lib(); L.fh_init(0)
hip = C.CDLL("libamdhip64.so")
det, rec = fa.FaceDetector(), fa.FaceRecognizer()
assert det.loadModel(models.cached("det_500m_seed100.onnx", models.make_det_500m)) and rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
B, F, K = 128, 1, 12
data = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
faces = [torch.zeros((B*F, 15), device="cuda") for _ in range(K)]
fo = [torch.zeros(B*F, dtype=torch.int32, device="cuda") for _ in range(K)]
emb = [torch.zeros((B*F, 512), device="cuda") for _ in range(K)]
tot = torch.zeros(K, dtype=torch.int32, device="cuda")
NCU = torch.cuda.get_device_properties(0).multi_processor_count

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[int(sum(1 << (i - 32 * w) for i in bits if 32 * w <= i < 32 * (w + 1))) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s.value

def run(sd, sr):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        fa.pipeline_submit_dev(det, rec, data.data_ptr(), B, 640, 640, F, faces[k].data_ptr(), fo[k].data_ptr(), emb[k].data_ptr(),
                               tot[k:].data_ptr(), sd, sr)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return int(tot.sum()) / dt, dt / K * 1e3

ref = torch.zeros((B*F, 512), device="cuda"); f0 = torch.zeros((B*F, 15), device="cuda"); o0 = torch.zeros(B*F, dtype=torch.int32, device="cuda")
n = fa.pipeline_run_dev(det, rec, data.data_ptr(), B, 640, 640, F, f0.data_ptr(), o0.data_ptr(), ref.data_ptr())
torch.cuda.synchronize()
def diff():
    return " ".join(f"{float((emb[k][:n] - ref[:n]).abs().max()):.1e}" for k in (0, K // 2, K - 1)) + f" faces_equal={bool(torch.equal(faces[K-1][:n], f0[:n]))}"
s0 = torch.cuda.Stream(); s1 = torch.cuda.Stream()
for name, (sd, sr) in (("one stream", (s0.cuda_stream, s0.cuda_stream)), ("two streams", (s0.cuda_stream, s1.cuda_stream))):
    run(sd, sr); v, ms = run(sd, sr)
    print(f"{name:28s} {v:8.1f} faces/s {ms:6.2f} ms/step   maxdiff {diff()}", flush=True)
for layout in ("low", "interleaved"):
    for nd in [int(a) for a in sys.argv[1:]] or [32, 48, 64]:
        if layout == "low":
            dbits = list(range(nd))
        else:                                   # nd/8 CUs out of every group of 32
            dbits = [g * 32 + i for g in range(8) for i in range(nd // 8)]
        rbits = [i for i in range(NCU) if i not in set(dbits)]
        sd, sr = masked_stream(dbits), masked_stream(rbits)
        L.fh_det_set_cus(det.handle, len(dbits)); L.fh_rec_set_cus(rec.handle, len(rbits))
        run(sd, sr); v, ms = run(sd, sr)
        print(f"mask {layout:11s} det {len(dbits):3d} rec {len(rbits):3d} {v:8.1f} faces/s {ms:6.2f} ms/step   maxdiff {diff()}", flush=True)
        hip.hipStreamDestroy(C.c_void_p(sd)); hip.hipStreamDestroy(C.c_void_p(sr))
