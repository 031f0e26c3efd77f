"""Phase timers of dwpw_reg_kernel (needs libfacehip built with -DFACEHIP_DWPW_PROF, see scripts/dwpw_prof.sh): runs ONE stride-1
depthwise->pointwise block of the given shape at batch 128 through a 3-conv graph and prints cycles per tile and phase."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth.onnx_writer import OnnxBuilder
H, W, Cc, Cout = (int(x) for x in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 128
DS = int(sys.argv[6]) if len(sys.argv) > 6 else 1                      # depthwise stride
L = fa.lib(); L.fh_init(0)
rng = np.random.default_rng(0)
b = OnnxBuilder("dwpw")
x = b.add_input("input", [1, 3, H, W])
def conv(x, w, bias, relu=True, **kw):
    y = b.node("Conv", [x, b.init(b.uid("w"), w.astype(np.float32)), b.init(b.uid("b"), bias.astype(np.float32))], **kw)
    return b.node("Relu", [y]) if relu else y
y = conv(x, rng.standard_normal((Cc, 3, 3, 3)) / 5, rng.standard_normal(Cc) / 10, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1])
y = conv(y, rng.standard_normal((Cc, 1, 3, 3)) / 3, rng.standard_normal(Cc) / 10, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[DS, DS], group=Cc)
y = conv(y, rng.standard_normal((Cout, Cc, 1, 1)) / np.sqrt(Cc), rng.standard_normal(Cout) / 10, kernel_shape=[1, 1], strides=[1, 1])
y = b.node("Transpose", [y], perm=[0, 2, 3, 1])
b.node("Reshape", [y, b.init("shape", np.array([-1, Cout], np.int64))], outputs=["out"])
b.add_output("out", ["A", Cout])
path = b.save("/tmp/dwpw_prof.onnx")
det = fa.FaceDetector(); assert det.loadModel(path)
data = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
run = lambda: L.fh_det_run_network_dev(det.handle, data.data_ptr(), B, H, W, W * 3, H * W * 3, 0)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L.fh_timing_enable(1)
for _ in range(5): run()
torch.cuda.synchronize()
cap = 1000; ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); tg = (C.c_int * cap)()
n = L.fh_timing_collect_ops(ms, fl, tg, cap); per = n // 5
print("per-op us:", [round(float(np.median([ms[r * per + i] for r in range(5)])) * 1e3, 1) for i in range(per)])
L.fh_timing_enable(0)
run(); torch.cuda.synchronize()
ws = L.fh_det_workspace_dev(det.handle)
nwg = 256 * 4 * 2
buf = np.zeros((nwg, 4, 8), np.int64)
assert L.fh_memcpy_d2h(buf.ctypes.data, ws, buf.nbytes) == 0
live = buf[buf[..., 6] > 0]
if len(live):
    tiles = live[:, 6].astype(np.float64)
    names = ["barrier A (prev tile's readers done + wait for prefetched loads)", "pf regs -> LDS", "issue next prefetch", "barrier B", "bias + K loop", "stores"]
    if Cc == 16 and Cout <= 32:     # this shape runs as front_kernel (stem inside the block): its own phase list
        names = ["window registers -> LDS (waits for the prefetched loads)", "issue next window prefetch", "barrier", "stem (bf16 MFMA, 3 terms)",
                 "barrier + depthwise -> pointwise (register-fed)", "stores"]
    tot = 0
    for i, nm in enumerate(names):
        c = (live[:, i] / tiles).mean(); tot += c
        print(f"{c:9.0f} cycles/tile  {nm}")
    print(f"{tot:9.0f} cycles/tile total; {tiles.mean():.1f} tiles per workgroup, {len(live) // 4} workgroups")
    wave_tot = live[:, :6].sum(axis=1).astype(np.float64)      # cycles a wave spent in its tile loop: the kernel ends with the SLOWEST one
    print(f"per-wave loop cycles: min {wave_tot.min():.0f}, median {np.median(wave_tot):.0f}, mean {wave_tot.mean():.0f}, max {wave_tot.max():.0f}  "
          f"(max / mean = {wave_tot.max() / wave_tot.mean():.2f})")
    # where the slow waves sit: per XCD (workgroup id % 8) and per position inside the XCD's run (id // 8)
    wg_tot = buf[:, :, :6].sum(axis=2).max(axis=1).astype(np.float64)     # slowest wave of each workgroup
    used = buf[:, 0, 6] > 0
    ids = np.nonzero(used)[0]
    if len(ids):
        per_xcd = [wg_tot[ids[ids % 8 == x]].mean() for x in range(8)]
        print("per-XCD mean of a workgroup's slowest wave:", " ".join(f"{v:.0f}" for v in per_xcd))
        pos = ids // 8
        q = np.quantile(pos, [0.25, 0.5, 0.75])
        parts = [wg_tot[ids[pos <= q[0]]].mean(), wg_tot[ids[(pos > q[0]) & (pos <= q[1])]].mean(), wg_tot[ids[(pos > q[1]) & (pos <= q[2])]].mean(), wg_tot[ids[pos > q[2]]].mean()]
        print("by position in the XCD's workgroup list (quartiles):", " ".join(f"{v:.0f}" for v in parts))
        order = np.argsort(wg_tot[ids])
        print("slowest 12 workgroups (id, cycles, tiles):", [(int(ids[k]), int(wg_tot[ids[k]]), int(buf[ids[k], 0, 6])) for k in order[-12:]])
        print("fastest 6 workgroups:", [(int(ids[k]), int(wg_tot[ids[k]]), int(buf[ids[k], 0, 6])) for k in order[:6]])
    mhz = live[:, 7] / 10.0                                     # shader cycles per 100 MHz tick x 100
    print(f"in-kernel shader clock (s_memtime / s_memrealtime over the tile loop): median {np.median(mhz):.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}")
else:
    print("no stamps: library not built with -DFACEHIP_DWPW_PROF")
