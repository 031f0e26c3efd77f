"""Single-layer timing of the fused F(2x2,3x3) kernel (conv_wino2.hip) against the direct kernels (tuning aid):
   python scripts/wino2_bench.py [B]      -> per shape: wino2 us (median of kernel-timer records), direct us"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = fa.lib(); L.fh_init(0)
rng = np.random.default_rng(0)
for (H, W, Cin, Cout) in ((56, 56, 64, 64), (112, 112, 64, 64), (56, 56, 64, 128), (80, 80, 64, 64)):
    x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
    w0 = (rng.standard_normal((Cout, 9, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = torch.zeros(Cout, device="cuda"); out = torch.zeros((B, H, W, Cout), device="cuda")
    L.fh_timing_enable(1)
    for _ in range(6):
        assert L.fh_conv_wino2_dev(x.data_ptr(), w0.ctypes.data, b.data_ptr(), 0, 0, out.data_ptr(), B, H, W, Cin, Cout, 0, 0, 0) == 0, fa._lib.last_error()
    cap = 1000
    ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); tg = (C.c_int * cap)()
    n = L.fh_timing_collect_ops(ms, fl, tg, cap)
    t2 = np.median([ms[i] for i in range(n) if tg[i] == 12][1:])
    rows, kpad = L.fh_conv_wt_rows(Cout), L.fh_conv_kpad(9 * Cin)
    wp = np.zeros((rows, kpad), np.float32)
    L.fh_conv_pack_weights(w0.ctypes.data, Cout, Cin, 3, wp.ctypes.data)
    wd = torch.from_numpy(wp).cuda()
    for _ in range(6):
        assert L.fh_conv_forward_dev(x.data_ptr(), wd.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, 3, 1, kpad, -1, 0) == 0
    n = L.fh_timing_collect_ops(ms, fl, tg, cap)
    per = n // 6
    td = sum(np.median([ms[r * per + i] for r in range(1, 6)]) for i in range(per))
    L.fh_timing_enable(0)
    direct = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"B={B} {H}x{W}x{Cin}->{Cout}: wino2 {t2*1e3:7.1f} us ({direct/2.25/t2/1e9:6.1f} TF/s executed, {direct/t2/1e9:6.1f} direct-equivalent)   "
          f"direct {td*1e3:7.1f} us ({direct/td/1e9:6.1f} TF/s)   x{td/t2:.2f}", flush=True)
