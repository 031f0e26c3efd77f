import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
L = fa.lib(); L.fh_init(0)
rec = fa.FaceRecognizer(); t0 = time.time(); assert rec.loadModel(models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)); print("load s", round(time.time() - t0, 2))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
crops = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (B, 112, 112, 3), dtype=np.uint8)).cuda()
res = {}
for on in (0, 1):
    L.fh_rec_set_winograd(rec.handle, on)
    out = torch.zeros((B, 512), device="cuda"); raw = torch.zeros((B, 512), device="cuda")
    for _ in range(3): rec.embed_aligned_dev(crops.data_ptr(), B, out.data_ptr(), raw.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): rec.embed_aligned_dev(crops.data_ptr(), B, out.data_ptr(), raw.data_ptr())
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    res[on] = (out.cpu().numpy().astype(np.float64), raw.cpu().numpy().astype(np.float64))
    print(f"winograd={on}: {ms:.2f} ms / {B} faces = {B/ms*1e3:.0f} faces/s", flush=True)
e0, r0 = res[0]; e1, r1 = res[1]
print("raw scale max %.3f  max abs diff %.2e  rel %.2e" % (np.abs(r0).max(), np.abs(r1 - r0).max(), np.abs(r1 - r0).max() / np.abs(r0).max()))
print("1 - cos max %.2e" % (1 - (e0 * e1).sum(1)).max())
