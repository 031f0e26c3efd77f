import sys, numpy as np, torch, time
sys.path.insert(0, "/root/repo")
import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
fa.lib().fh_init(0)
B=16
host = np.random.default_rng(0).integers(0,256,(B,640,640,3),dtype=np.uint8)
data = torch.from_numpy(host).cuda()
for bias in (-3.5,-3.0,-2.5):
    p = models.make_det_500m(f"/tmp/det_{bias}.onnx", cls_bias=bias)
    det = fa.FaceDetector(); assert det.loadModel(p)
    faces = torch.zeros((B*512,15),device="cuda"); counts = torch.zeros(B,dtype=torch.int32,device="cuda")
    for thr in (0.5,0.3):
        det.detect_batch_dev(data.data_ptr(), B, 640, 640, faces.data_ptr(), 512, counts.data_ptr(), thr, 0.4)
        torch.cuda.synchronize()
        # raw score stats
        print(bias, thr, counts.cpu().numpy())
    import ctypes as C
    r,c=C.c_int(),C.c_int()
    ptr = fa.lib().fh_det_output_dev(det.handle,0,C.byref(r),C.byref(c))
    s=np.empty((B,r.value,c.value),np.float32); fa.lib().fh_memcpy_d2h(s.ctypes.data, ptr, s.nbytes)
    lg=np.log(s/(1-s)); print("  logit mean/std", lg.mean(), lg.std(), "max score", s.max())
