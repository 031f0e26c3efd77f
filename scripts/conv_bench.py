"""Single conv-layer micro-benchmark (tuning aid):
   python scripts/conv_bench.py B H W Cin Cout k stride [cfg] [iters]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
a = [int(x) for x in sys.argv[1:]]
B, H, W, Cin, Cout, k, stride = a[:7]
cfg = a[7] if len(a) > 7 else -1
iters = a[8] if len(a) > 8 else 20
L = fa.lib(); L.fh_init(0)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
rows, kpad = L.fh_conv_wt_rows(Cout), L.fh_conv_kpad(k * k * Cin)
wp = np.zeros((rows, kpad), np.float32); w0 = (rng.standard_normal((Cout, k * k, Cin)) / np.sqrt(k * k * Cin)).astype(np.float32)
L.fh_conv_pack_weights(w0.ctypes.data, Cout, Cin, k, wp.ctypes.data)
w = torch.from_numpy(wp).cuda(); b = torch.zeros(Cout, device="cuda")
Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
out = torch.zeros((B, Ho, Wo, Cout), device="cuda")
run = lambda: L.fh_conv_forward_dev(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, k, stride, kpad, cfg, 0)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
print(f"B={B} {H}x{W}x{Cin}->{Cout} k{k}s{stride} cfg{cfg}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s  ({fl/ms/1e9/157.3*100:.1f}% of f32 MFMA peak)")
