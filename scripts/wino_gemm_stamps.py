"""Phase stamps of wino_gemm_kernel on one Winograd layer (diagnostic build -DFACEHIP_WINO_STAMP loaded through FACEHIP_LIB, see
scripts/wino_gemm_stamps.sh): python scripts/wino_gemm_stamps.py B H W Cin Cout"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
B, H, W, Cin, Cout = (int(x) for x in sys.argv[1:6]) if len(sys.argv) >= 6 else (128, 14, 14, 256, 256)
L = fa.lib(); L.fh_init(0)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).cuda()
w = np.ascontiguousarray((rng.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32))
b = torch.zeros(Cout, device="cuda"); out = torch.zeros((B, H, W, Cout), device="cuda")
for _ in range(4):                                                     # (the entry point is synchronous; the last launch's stamps stay)
    assert L.fh_conv_winograd_dev(x.data_ptr(), w.ctypes.data, b.data_ptr(), out.data_ptr(), B, H, W, Cin, Cout, None) == 0
fn = L.fh_debug_wino_stamps
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((4096, 8), np.uint64)
assert fn(buf.ctypes.data, buf.size) == 0
t = buf[buf[:, 4] > 0][:, :5].astype(np.int64)
xcc = buf[buf[:, 4] > 0][:, 6].astype(np.int64)
t0 = t[:, 0].min()
us = lambda a: a / 100.0
print(f"== B={B} {H}x{W}x{Cin}->{Cout}: {len(t)} workgroups (first 4096 of the launch); span first entry -> last acknowledgement {us(t[:, 4].max() - t0):.1f} us")
rows = (("entry -> first chunk landed", t[:, 1] - t[:, 0]), ("K loop", t[:, 2] - t[:, 1]), ("K loop done -> last store issued", t[:, 3] - t[:, 2]),
        ("last store issued -> acknowledged", t[:, 4] - t[:, 3]), ("whole workgroup", t[:, 4] - t[:, 0]))
for name, v in rows:
    v = us(v)
    print(f"  {name:38s} median {np.median(v):6.2f} us   p10 {np.percentile(v, 10):6.2f}   p90 {np.percentile(v, 90):6.2f}   max {v.max():6.2f}")
start = us(t[:, 0] - t0)
print("  workgroup start times (us), deciles:", np.round(np.percentile(start, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]), 1))
end = us(t[:, 4] - t0)
print("  workgroup end times (us), deciles:  ", np.round(np.percentile(end, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100]), 1))
first = start < np.percentile(start, 50)
for name, m in (("first-round workgroups (start < median)", first), ("later workgroups", ~first)):
    if m.any():
        print(f"  {name}: K loop median {np.median(us(t[m, 2] - t[m, 1])):.2f} us, prologue {np.median(us(t[m, 1] - t[m, 0])):.2f}, whole {np.median(us(t[m, 4] - t[m, 0])):.2f}")
print("  workgroups per XCC id:", np.bincount(xcc, minlength=8).tolist())
