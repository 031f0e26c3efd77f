"""Phase stamps of conv_pw_kernel on SCRFD's 20x20x288 layers (diagnostic build -DFACEHIP_PW_ABL loaded through FACEHIP_LIB, see
scripts/pw_ablate.sh): per workgroup, 100 MHz ticks at kernel entry / first chunk landed / K loop done / epilogue done + stores acknowledged."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
B = 128
det = fa.FaceDetector(); assert det.loadModel(models.cached("det_500m_seed100.onnx", models.make_det_500m))
frames = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda")
L = fa.lib()
for _ in range(3):
    assert L.fh_det_run_network_dev(det.handle, frames.data_ptr(), B, 640, 640, 1920, 640 * 640 * 3, None) == B
torch.cuda.synchronize()
ws = L.fh_det_workspace_dev(det.handle)
n = 1200
buf = np.zeros((n, 5), np.uint64)
assert L.fh_memcpy_d2h(buf.ctypes.data, ws, buf.nbytes) == 0
buf = buf[buf[:, 4] == 1]
t = buf[:, :4].astype(np.int64)
t0 = t[:, 0].min()
pro, loop, epi = (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t[:, 1]) / 100.0, (t[:, 3] - t[:, 2]) / 100.0
print(f"{len(t)} workgroups; kernel span {(t[:, 3].max() - t0) / 100.0:.1f} us")
for name, v in (("entry -> first chunk landed", pro), ("K loop", loop), ("epilogue (incl. store acknowledgement)", epi)):
    print(f"  {name:42s} median {np.median(v):7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}   max {v.max():7.2f}")
b2 = np.zeros((8192 + n * 4,), np.uint64)
assert L.fh_memcpy_d2h(b2.ctypes.data, ws, b2.nbytes) == 0
e = b2[8192:].reshape(n, 4).astype(np.int64)
ok = e[:, 0] > 0
if ok.any():
    k2 = buf[:, 2].astype(np.int64) if False else t[:, 2]
    for name, v in (("K loop done -> ep vectors parked + barrier", (e[ok, 3] - t[ok, 2]) / 100.0), ("barrier -> epilogue entry", (e[ok, 0] - e[ok, 3]) / 100.0),
                    ("entry -> first block in LDS", (e[ok, 1] - e[ok, 0]) / 100.0), ("first block in LDS -> last store issued", (e[ok, 2] - e[ok, 1]) / 100.0)):
        print(f"  {name:42s} median {np.median(v):7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}   max {v.max():7.2f}")
b3 = np.zeros((16384 + n * 2,), np.uint64)
assert L.fh_memcpy_d2h(b3.ctypes.data, ws, b3.nbytes) == 0
e2 = b3[16384:].reshape(n, 2).astype(np.int64)
if ok.any() and (e2[ok, 0] > 0).all():
    for name, v in (("first block: accumulators in LDS -> its reads landed", (e2[ok, 0] - e[ok, 1]) / 100.0), ("first block: reads landed -> its 4 stores issued", (e2[ok, 1] - e2[ok, 0]) / 100.0)):
        print(f"  {name:52s} median {np.median(v):7.2f} us   p10 {np.percentile(v, 10):7.2f}   p90 {np.percentile(v, 90):7.2f}   max {v.max():7.2f}")
start = (t[:, 0] - t0) / 100.0
order = np.argsort(start)
print("  start times (us) of the workgroups, deciles:", np.round(np.percentile(start, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]), 1))
