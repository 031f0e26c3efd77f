"""Phase stamps of wino2_kernel (needs the diagnostic library: scripts/wino2_prof.sh): one layer at batch B, cycles per workgroup phase.
   python scripts/wino2_prof.py H W Cin Cout [B]"""
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
ws = L.fh_debug_wino2_stamps()
L.fh_timing_enable(1)
for _ in range(3):
    assert L.fh_conv_wino2_dev(x.data_ptr(), w0.ctypes.data, b.data_ptr(), 0, 0, out.data_ptr(), B, H, W, Cin, Cout, 0, 0, 0) == 0, fa._lib.last_error()
torch.cuda.synchronize()
cap = 100; ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); tg = (C.c_int * cap)()
nrec = L.fh_timing_collect_ops(ms, fl, tg, cap)
print("kernel time of the stamped build:", [round(ms[i] * 1e3, 1) for i in range(nrec) if tg[i] == 12], "us")
L.fh_timing_enable(0)
buf = np.zeros((4096, 4, 8), np.int64)
assert L.fh_memcpy_d2h(buf.ctypes.data, ws, buf.nbytes) == 0
live = buf[buf[..., 6] > 0]
if not len(live):
    print("no stamps: library not built with -DFACEHIP_W2_PROF"); sys.exit(0)
n = 8
# stamps (cumulative since the wave's start): [0] prologue done, [5] K loop (2 halo loads + 32 stages) + vectors to LDS done, [6] epilogue done
# (stamps INSIDE the K loop are not taken: a cycle-counter read there is a scheduling barrier that costs the loop ~450 spilled registers)
names = {0: "prologue (index arithmetic, vectors)", 5: "K loop: halo loads + 32 stages (+ barriers, vectors to LDS)", 6: "epilogue (residual, stores)"}
prev = np.zeros(len(live))
for i in (0, 5, 6):
    cur = live[:, i].astype(np.float64)
    print(f"{(cur - prev).mean():9.0f} cycles (min {(cur - prev).min():7.0f} max {(cur - prev).max():7.0f})  {names[i]}")
    prev = cur
print(f"{prev.mean():9.0f} cycles per wave in all; {len(live) // 4} workgroups stamped; shader clock median {np.median(live[:, 7]) / 10:.0f} MHz")
print("MFMA issue floor of one wave's K loop: 32 stages x 16 x 64 = 32768 cycles per workgroup (x2 workgroups per CU share each SIMD)")
