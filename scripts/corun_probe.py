"""Does a bandwidth-bound kernel run 'for free' beside the MFMA-bound recogniser? (two streams)"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
L = fa.lib(); L.fh_init(0)
rec = fa.FaceRecognizer(); assert rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50))
det = fa.FaceDetector(); assert det.loadModel(models.cached("det_500m_seed100.onnx", models.make_det_500m))
B = 128
crops = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 112, 112, 3), dtype=np.uint8)).cuda()
frames = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
out = torch.zeros((B, 512), device="cuda")
x = torch.empty(256 << 20, dtype=torch.float32, device="cuda").normal_(); y = torch.empty_like(x)     # 1 GiB each
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 10
def rec_loop():
    for _ in range(N): L.fh_rec_embed_aligned_dev(rec.handle, crops.data_ptr(), B, out.data_ptr(), 0, s1.cuda_stream)
def copy_loop(n):
    with torch.cuda.stream(s2):
        for _ in range(n): y.copy_(x)
def det_loop(n):
    for _ in range(n): L.fh_det_run_network_dev(det.handle, frames.data_ptr(), B, 640, 640, 1920, 640 * 1920, s2.cuda_stream)
def timed(f):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
t_rec = timed(rec_loop)
print(f"rec alone          {t_rec/N:7.2f} ms/batch")
t_copy = timed(lambda: copy_loop(40))
print(f"copy alone         {t_copy/40:7.3f} ms per 2 GiB moved ({2*1.0737/ (t_copy/40/1e3)/1e3:.2f} TB/s)")
t_det = timed(lambda: det_loop(N))
print(f"det alone          {t_det/N:7.2f} ms/batch")
ncopy = int(t_rec / (t_copy / 40))
t_both = timed(lambda: (rec_loop(), copy_loop(ncopy)))
print(f"rec + {ncopy} copies   {t_both:7.2f} ms  (sum alone {t_rec + ncopy * t_copy / 40:.2f}, max alone {max(t_rec, ncopy * t_copy / 40):.2f})")
t_both = timed(lambda: (rec_loop(), det_loop(N)))
print(f"rec + det          {t_both/N:7.2f} ms/batch (sum alone {(t_rec + t_det)/N:.2f}, max alone {max(t_rec, t_det)/N:.2f})")
