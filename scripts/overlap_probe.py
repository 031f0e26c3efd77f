import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo" if len(sys.argv) < 2 else sys.argv[1])
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
fa.lib().fh_init(0)
det, rec = fa.FaceDetector(), fa.FaceRecognizer()
assert det.loadModel(models.cached("det_500m_seed100.onnx", models.make_det_500m)) and rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
B, F, K = 128, 1, 12
data = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
faces = [torch.zeros((B*F, 15), device="cuda") for _ in range(K)]
fo = [torch.zeros(B*F, dtype=torch.int32, device="cuda") for _ in range(K)]
emb = [torch.zeros((B*F, 512), device="cuda") for _ in range(K)]
tot = torch.zeros(K, dtype=torch.int32, device="cuda")
import os
pr = int(os.environ.get('DET_PRIO', '0'))
sd, sr = torch.cuda.Stream(priority=pr), torch.cuda.Stream(priority=int(os.environ.get('REC_PRIO', '0')))
def run(two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K):
        fa.pipeline_submit_dev(det, rec, data.data_ptr(), B, 640, 640, F, faces[k].data_ptr(), fo[k].data_ptr(), emb[k].data_ptr(),
                               tot[k:].data_ptr(), sd.cuda_stream, (sr if two else sd).cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return int(tot.sum()) / dt, dt / K * 1e3
for two in (0, 1, 0, 1):
    run(two)
    v, ms = run(two)
    print("two streams" if two else "one stream ", round(v, 1), "faces/s", round(ms, 2), "ms/step")
ref = torch.zeros((B*F, 512), device="cuda"); f0 = torch.zeros((B*F,15),device="cuda"); o0=torch.zeros(B*F,dtype=torch.int32,device="cuda")
n = fa.pipeline_run_dev(det, rec, data.data_ptr(), B, 640, 640, F, f0.data_ptr(), o0.data_ptr(), ref.data_ptr())
torch.cuda.synchronize()
print("same results:", n, bool(torch.equal(ref[:n], emb[K-1][:n])), bool(torch.equal(ref[:n], emb[0][:n])))
