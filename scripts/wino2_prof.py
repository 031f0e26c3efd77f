"""Kernel time of wino2_kernel on one layer at batch B under the ablation switches of the diagnostic build (scripts/wino2_prof.sh sets
FACEHIP_LIB / FACEHIP_W2_ABLATE):   python scripts/wino2_prof.py H W Cin Cout [B]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
H, W, Cin, Cout = (int(x) for x in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 128
L = fa.lib(); L.fh_init(0)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
w0 = (rng.standard_normal((Cout, 9, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
b = torch.zeros(Cout, device="cuda"); out = torch.zeros((B, H, W, Cout), device="cuda")
L.fh_timing_enable(1)
for _ in range(4):
    assert L.fh_conv_wino2_dev(x.data_ptr(), w0.ctypes.data, b.data_ptr(), 0, 0, out.data_ptr(), B, H, W, Cin, Cout, 0, 0, 0) == 0, fa._lib.last_error()
torch.cuda.synchronize()
cap = 100; ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); tg = (C.c_int * cap)()
nrec = L.fh_timing_collect_ops(ms, fl, tg, cap)
print("kernel time:", [round(ms[i] * 1e3, 1) for i in range(nrec) if tg[i] == 12][1:], "us;  in-kernel shader clock (median over the waves of the last launch):",
      round(L.fh_debug_wino2_clock_mhz()), "MHz")
