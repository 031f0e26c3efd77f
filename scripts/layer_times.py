"""Per-layer timing of one network (tuning aid): python scripts/layer_times.py rec|det [batch] [cfg]"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
which = sys.argv[1] if len(sys.argv) > 1 else "rec"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if which == "rec" else 128)
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else -1
sk = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L = fa.lib(); L.fh_init(0)
rng = np.random.default_rng(0)
if which == "rec":
    path = models.cached("w600k_r50_seed200.onnx", models.make_w600k_r50)
    net = fa.FaceRecognizer(); assert net.loadModel(path)
    L.fh_rec_set_conv_cfg(net.handle, cfg, sk)
    data = torch.from_numpy(rng.integers(0, 256, (B, 112, 112, 3), dtype=np.uint8)).cuda()
    out = torch.zeros((B, 512), device="cuda")
    run = lambda: net.embed_aligned_dev(data.data_ptr(), B, out.data_ptr())
    desc = fa.plan_describe(path, 112, 112)
else:
    path = models.cached("det_500m_seed100.onnx", models.make_det_500m)
    net = fa.FaceDetector(); assert net.loadModel(path)
    L.fh_det_set_conv_cfg(net.handle, cfg, sk)
    front = os.environ.get("FACEHIP_NO_FRONT") != "1"
    L.fh_det_set_fused_front(net.handle, 1 if front else 0)
    data = torch.from_numpy(rng.integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)).cuda()
    run = lambda: L.fh_det_run_network_dev(net.handle, data.data_ptr(), B, 640, 640, 1920, 640 * 1920, 0)
    desc = fa.plan_describe(path, 640, 640)
import re
ops = [l for l in desc.splitlines()[1:] if l and l[0].isdigit()]
folded = {int(m.group(1)) for l in ops for m in [re.search(r"sc<-op(\d+)", l)] if m}      # shortcuts that run inside their consumer
if os.environ.get("FACEHIP_NO_SC_FOLD") != "1":
    ops = [l for l in ops if int(l.split()[0]) not in folded]
if which == "det" and front:                                     # ops 0 + 1 run as one kernel (stem inside the first depthwise block)
    ops = ["0+1 fused front: " + ops[1].split(" ", 1)[1]] + ops[2:]
for _ in range(3): run()
torch.cuda.synchronize()
L.fh_timing_enable(1)
R = 5
for _ in range(R): run()
torch.cuda.synchronize()
cap = 100000
ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); tg = (C.c_int * cap)()
n = L.fh_timing_collect_ops(ms, fl, tg, cap)
per = n // R
tot = 0
oi = 0
for i in range(per):
    t = np.median([ms[r * per + i] for r in range(R)])
    tot += t
    if tg[i] in (6, 8):
        print(f"{t*1e3:9.1f} us             {'fix-up' if tg[i] == 6 else 'winograd transform'}")
        continue
    if tg[i] == 10:                                              # whole tile rounds of the next op in conv_tall_kernel; its remainder follows
        desc = ops[oi] if oi < len(ops) else ''
        mm = re.search(r"MMAC ([0-9.]+)", desc)
        whole = bool(mm) and fl[i] >= 0.999 * 2e6 * float(mm.group(1)) * B      # conv_tall_kernel took every tile: no remainder launch follows
        print(f"{t*1e3:9.1f} us  {fl[i]/t/1e9 if t>0 else 0:7.1f} TF/s  tall   {desc}  ({'all tiles' if whole else 'whole tile rounds'})")
        if whole: oi += 1
        continue
    print(f"{t*1e3:9.1f} us  {fl[i]/t/1e9 if t>0 else 0:7.1f} TF/s  cfg{tg[i]}  {ops[oi] if oi < len(ops) else ''}")
    oi += 1
print(f"total {tot:.3f} ms for batch {B}; {sum(fl[i] for i in range(per))/tot/1e9:.1f} TF/s overall")
