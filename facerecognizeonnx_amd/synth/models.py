"""Seeded synthetic ONNX models with the public InsightFace architectures.

* IResNet (ArcFace ``w600k_r50`` = arcface_torch ``iresnet50``) — SURVEY.md A.1.
  Op set: Conv, BatchNormalization, PRelu, Add, Flatten, Gemm.
* SCRFD ``det_500m`` = ``scrfd_500m_bnkps`` (MobileNetV1 backbone, PAFPN, per-stride
  depthwise-separable heads, 9 outputs) — SURVEY.md A.2.
  Op set: Conv (dense + depthwise), Relu, Add, Resize(nearest), Sigmoid, Transpose, Reshape.
* A "pre-decoded" detector whose single output is ``[1, N, 15]`` — the only layout the
  reference's postprocess understands literally (reference src/face_detector.cpp:242-278).

Weights follow SURVEY.md §8(d): conv ~ N(0, g/fan_in), BN gamma~U(.5,1.5), beta~N(0,.1),
mean~N(0,.1), var~U(.5,1.5), PReLU slope .25*U(.5,1.5).  The genuine files are not
available offline; these graphs exercise the same loader / planner / kernels.
"""
from __future__ import annotations

import os
from typing import Sequence

import numpy as np

from .onnx_writer import OnnxBuilder

_BN_EPS = 1e-5


class _W:
    """Weight factory on one seeded generator."""

    def __init__(self, seed: int):
        self.rng = np.random.default_rng(seed)

    def conv(self, cout, cin_g, k, gain=2.0):
        fan_in = cin_g * k * k
        return (self.rng.standard_normal((cout, cin_g, k, k)) * np.sqrt(gain / fan_in)).astype(np.float32)

    def bias(self, c, std=0.05, mean=0.0):
        return (mean + self.rng.standard_normal(c) * std).astype(np.float32)

    def bn(self, c):
        g = self.rng.uniform(0.5, 1.5, c).astype(np.float32)
        b = (self.rng.standard_normal(c) * 0.1).astype(np.float32)
        m = (self.rng.standard_normal(c) * 0.1).astype(np.float32)
        v = self.rng.uniform(0.5, 1.5, c).astype(np.float32)
        return g, b, m, v

    def slope(self, c):
        return (0.25 * self.rng.uniform(0.5, 1.5, c)).astype(np.float32)


def _fold(w, b, bn):
    """Fold y = BN(conv(x)) into conv weights (what the public exports ship)."""
    g, beta, m, v = bn
    s = (g.astype(np.float64) / np.sqrt(v.astype(np.float64) + _BN_EPS))
    w2 = (w.astype(np.float64) * s[:, None, None, None]).astype(np.float32)
    b0 = np.zeros_like(s) if b is None else b.astype(np.float64)
    b2 = ((b0 - m) * s + beta).astype(np.float32)
    return w2, b2


# ----------------------------------------------------------------------------- IResNet
def make_iresnet(path: str, layers: Sequence[int] = (3, 4, 14, 3),
                 widths: Sequence[int] = (64, 128, 256, 512), size: int = 112,
                 feat: int = 512, seed: int = 200, fold_bn: bool = True,
                 batch_dim="N", downsample_first: bool = False, stage_strides: Sequence[int] = (2, 2, 2, 2)) -> str:
    """IResNet as arcface_torch builds it.

    fold_bn=True ships Conv(+bias) where a BN follows a conv (as the public file does);
    fold_bn=False keeps every BatchNormalization node so the loader's Conv→BN folding is
    exercised as well.  The pre-conv ``bn1`` of each block and the tail BNs always stay.
    downsample_first=True writes a block's shortcut convolution in front of its bn1 / conv1 nodes (a legal
    topological order some exporters produce): the block input then has its last listed reader BEFORE conv1.
    stage_strides: stride of each stage's first block (arcface_torch: 2 everywhere); 1 gives a stride-1 block with a 1x1 PROJECTION
    shortcut — a shape torchvision-style ResNets have and the engine's shortcut folding must not mistake for the strided one.
    """
    W = _W(seed)
    b = OnnxBuilder("iresnet")
    x = b.add_input("input.1", [batch_dim, 3, size, size])

    def bn_node(x, c, tag):
        g, beta, m, v = W.bn(c)
        n = [b.init(f"{tag}.weight", g), b.init(f"{tag}.bias", beta),
             b.init(f"{tag}.running_mean", m), b.init(f"{tag}.running_var", v)]
        return b.node("BatchNormalization", [x] + n, epsilon=float(_BN_EPS), momentum=0.9)

    def conv_bn(x, cin, cout, k, stride, tag, gain):
        w = W.conv(cout, cin, k, gain)
        bn = W.bn(cout)
        pad = k // 2
        if fold_bn:
            w2, b2 = _fold(w, None, bn)
            return b.node("Conv", [x, b.init(f"{tag}.weight", w2), b.init(f"{tag}.bias", b2)],
                          kernel_shape=[k, k], strides=[stride, stride], pads=[pad] * 4,
                          dilations=[1, 1], group=1)
        y = b.node("Conv", [x, b.init(f"{tag}.weight", w)],
                   kernel_shape=[k, k], strides=[stride, stride], pads=[pad] * 4,
                   dilations=[1, 1], group=1)
        g, beta, m, v = bn
        n = [b.init(f"{tag}.bn.weight", g), b.init(f"{tag}.bn.bias", beta),
             b.init(f"{tag}.bn.running_mean", m), b.init(f"{tag}.bn.running_var", v)]
        return b.node("BatchNormalization", [y] + n, epsilon=float(_BN_EPS), momentum=0.9)

    def prelu(x, c, tag):
        return b.node("PRelu", [x, b.init(f"{tag}.slope", W.slope(c).reshape(c, 1, 1))])

    x = conv_bn(x, 3, widths[0], 3, 1, "conv1", 2.0)
    x = prelu(x, widths[0], "prelu")
    cin = widths[0]
    for li, (nblk, planes) in enumerate(zip(layers, widths)):
        for bi in range(nblk):
            stride = stage_strides[li] if bi == 0 else 1
            tag = f"layer{li + 1}.{bi}"
            sc = None
            if bi == 0 and downsample_first:
                sc = conv_bn(x, cin, planes, 1, stride, f"{tag}.downsample", 1.0)
            y = bn_node(x, cin, f"{tag}.bn1")
            y = conv_bn(y, cin, planes, 3, 1, f"{tag}.conv1", 1.0)
            y = prelu(y, planes, f"{tag}.prelu")
            y = conv_bn(y, planes, planes, 3, stride, f"{tag}.conv2", 0.5)
            if bi == 0 and sc is None:
                sc = conv_bn(x, cin, planes, 1, stride, f"{tag}.downsample", 1.0)
            elif sc is None:
                sc = x
            x = b.node("Add", [y, sc])
            cin = planes
    x = bn_node(x, cin, "bn2")
    x = b.node("Flatten", [x], axis=1)
    sp = size
    for st in stage_strides:
        sp = (sp - 1) // st + 1
    kdim = cin * sp * sp
    wfc = (W.rng.standard_normal((feat, kdim)) * np.sqrt(1.0 / kdim)).astype(np.float32)
    x = b.node("Gemm", [x, b.init("fc.weight", wfc), b.init("fc.bias", W.bias(feat))],
               alpha=1.0, beta=1.0, transB=1)
    g, beta, m, v = W.bn(feat)
    n = [b.init("features.weight", g), b.init("features.bias", beta),
         b.init("features.running_mean", m), b.init("features.running_var", v)]
    out = b.node("BatchNormalization", [x] + n, outputs=["683"], epsilon=float(_BN_EPS), momentum=0.9)
    b.add_output(out, [batch_dim, feat])
    return b.save(path)


def make_mobilefacenet(path: str, blocks: Sequence[int] = (1, 4, 6, 2), base: int = 128, size: int = 112, feat: int = 512,
                       seed: int = 300, fold_bn: bool = True, batch_dim="N") -> str:
    """MobileFaceNet as arcface_torch builds it (``mbf``, scale 2 -> base = 128 channels): the ``w600k_mbf`` recogniser of
    buffalo_s / buffalo_sc ("keep the graph executor generic so w600k_mbf also loads").

    stem 3x3 s2 -> a GROUPED 3x3 (base/2 groups of 2 channels) -> three stages of [DepthWise s2, Residual x n]
    (DepthWise = 1x1 expand + PReLU, depthwise 3x3 + PReLU, 1x1 linear project) -> 1x1 conv_sep to 4*base channels
    -> GDC: depthwise k x k VALID conv over the whole map (k = size/16), Flatten, bias-less Linear (a MatMul node), BN.
    Op set: Conv (dense, grouped, depthwise 3x3, depthwise global), BatchNormalization, PRelu, Add, Flatten, MatMul.
    """
    W = _W(seed)
    b = OnnxBuilder("mobilefacenet")
    x = b.add_input("input.1", [batch_dim, 3, size, size])

    def conv_bn(x, cin, cout, k, stride, pad, groups, tag, gain=2.0):
        w = W.conv(cout, cin // groups, k, gain)
        bn = W.bn(cout)
        kw = dict(kernel_shape=[k, k], strides=[stride, stride], pads=[pad] * 4, dilations=[1, 1], group=groups)
        if fold_bn:
            w2, b2 = _fold(w, None, bn)
            return b.node("Conv", [x, b.init(f"{tag}.weight", w2), b.init(f"{tag}.bias", b2)], **kw)
        y = b.node("Conv", [x, b.init(f"{tag}.weight", w)], **kw)
        g, beta, m, v = bn
        n = [b.init(f"{tag}.bn.weight", g), b.init(f"{tag}.bn.bias", beta), b.init(f"{tag}.bn.running_mean", m),
             b.init(f"{tag}.bn.running_var", v)]
        return b.node("BatchNormalization", [y] + n, epsilon=float(_BN_EPS), momentum=0.9)

    def conv_block(x, cin, cout, k, stride, pad, groups, tag):          # Conv + BN + PReLU
        y = conv_bn(x, cin, cout, k, stride, pad, groups, tag)
        return b.node("PRelu", [y, b.init(f"{tag}.prelu", W.slope(cout).reshape(cout, 1, 1))])

    def depth_wise(x, cin, cout, stride, groups, residual, tag):
        y = conv_block(x, cin, groups, 1, 1, 0, 1, f"{tag}.conv")
        y = conv_block(y, groups, groups, 3, stride, 1, groups, f"{tag}.conv_dw")
        y = conv_bn(y, groups, cout, 1, 1, 0, 1, f"{tag}.project", gain=0.5 if residual else 1.0)
        return b.node("Add", [x, y]) if residual else y

    c1, c2 = base, 2 * base
    x = conv_block(x, 3, c1, 3, 2, 1, 1, "layers.0")
    if blocks[0] == 1:
        x = conv_block(x, c1, c1, 3, 1, 1, c1 // 2, "layers.1")         # groups = 64 at scale 2: two channels per group
    else:
        for i in range(blocks[0]):
            x = depth_wise(x, c1, c1, 1, c1, True, f"layers.1.{i}")
    x = depth_wise(x, c1, c1, 2, c1, False, "layers.2")
    for i in range(blocks[1]):
        x = depth_wise(x, c1, c1, 1, c1, True, f"layers.3.{i}")
    x = depth_wise(x, c1, c2, 2, 2 * c1, False, "layers.4")
    for i in range(blocks[2]):
        x = depth_wise(x, c2, c2, 1, 2 * c1, True, f"layers.5.{i}")
    x = depth_wise(x, c2, c2, 2, 4 * c1, False, "layers.6")
    for i in range(blocks[3]):
        x = depth_wise(x, c2, c2, 1, 2 * c1, True, f"layers.7.{i}")
    cs = 4 * base
    x = conv_block(x, c2, cs, 1, 1, 0, 1, "conv_sep")
    k = size // 16
    x = conv_bn(x, cs, cs, k, 1, 0, cs, "features.gdc", gain=1.0)       # global depthwise conv: [N, cs, 1, 1]
    x = b.node("Flatten", [x], axis=1)
    wfc = (W.rng.standard_normal((cs, feat)) * np.sqrt(1.0 / cs)).astype(np.float32)      # MatMul: [K, N]
    x = b.node("MatMul", [x, b.init("features.linear.weight_t", wfc)])
    g, beta, m, v = W.bn(feat)
    n = [b.init("features.bn.weight", g), b.init("features.bn.bias", beta), b.init("features.bn.running_mean", m),
         b.init("features.bn.running_var", v)]
    out = b.node("BatchNormalization", [x] + n, outputs=["516"], epsilon=float(_BN_EPS), momentum=0.9)
    b.add_output(out, [batch_dim, feat])
    return b.save(path)


def make_w600k_mbf(path: str, seed: int = 300) -> str:
    """The buffalo_s / buffalo_sc recogniser: MobileFaceNet, blocks (1, 4, 6, 2), scale 2, 112x112 -> 512-d."""
    return make_mobilefacenet(path, seed=seed)


def make_w600k_r50(path: str, seed: int = 200) -> str:
    """Full-size IResNet-50 (43.6 M params, 6.31 GMAC/face)."""
    return make_iresnet(path, (3, 4, 14, 3), (64, 128, 256, 512), 112, 512, seed)


# ----------------------------------------------------------------------------- SCRFD
def make_scrfd(path: str, stage_blocks: Sequence[int] = (2, 3, 2, 6),
               stage_planes: Sequence[int] = (16, 16, 40, 72, 152, 288),
               neck_ch: int = 16, head_ch: int = 64, seed: int = 100,
               cls_bias: float = -3.0, static_hw: int | None = None, cls_gain: float = 20.0,
               dynamic_resize: bool = False) -> str:
    """scrfd_500m_bnkps topology; BN folded as in the public export.

    ``cls_gain`` / ``cls_bias`` give score logits ~ N(cls_bias, 1) on random frames so that only a
    realistic fraction (~0.1 %) of the 16 800 anchors clears the 0.5 threshold; bbox distances are biased positive
    so decoded boxes are proper rectangles and NMS has real overlaps to resolve.
    """
    W = _W(seed)
    b = OnnxBuilder("scrfd")
    hw = static_hw if static_hw else "?"
    x = b.add_input("input.1", [1, 3, hw, hw])

    def conv(x, cin, cout, k, stride, tag, relu, group=1, gain=2.0, bias_mean=0.0, bias_std=0.05):
        w = W.conv(cout, cin // group, k, gain)
        y = b.node("Conv", [x, b.init(f"{tag}.weight", w),
                            b.init(f"{tag}.bias", W.bias(cout, bias_std, bias_mean))],
                   kernel_shape=[k, k], strides=[stride, stride], pads=[k // 2] * 4,
                   dilations=[1, 1], group=group)
        return b.node("Relu", [y]) if relu else y

    def conv_dw(x, cin, cout, stride, tag):
        x = conv(x, cin, cin, 3, stride, f"{tag}.dw", True, group=cin)
        return conv(x, cin, cout, 1, 1, f"{tag}.pw", True)

    p = stage_planes
    x = conv(x, 3, p[0], 3, 2, "backbone.stem.0", True)
    x = conv_dw(x, p[0], p[1], 1, "backbone.stem.1")
    cin = p[1]
    feats = []
    for si, nb in enumerate(stage_blocks):
        for bi in range(nb):
            x = conv_dw(x, cin, p[si + 2], 2 if bi == 0 else 1, f"backbone.layer{si + 1}.{bi}")
            cin = p[si + 2]
        feats.append((x, cin))
    feats = feats[1:]                                  # strides 8, 16, 32

    lat = [conv(f, c, neck_ch, 1, 1, f"neck.lateral_convs.{i}", False, gain=1.0)
           for i, (f, c) in enumerate(feats)]
    scales = b.init("neck.up_scales", np.array([1, 1, 2, 2], np.float32))
    roi = b.init("neck.up_roi", np.zeros(0, np.float32))
    for i in (2, 1):
        if dynamic_resize:
            # what a dynamic-axes export of F.interpolate(x, size=prev.shape[2:]) looks like:
            # Shape -> Slice -> Concat -> Resize(sizes); the loader must fold it to a constant
            c0, c2, c4, ax0 = (b.init(b.uid("k"), np.array([v], np.int64)) for v in (0, 2, 4, 0))
            hw_ = b.node("Slice", [b.node("Shape", [lat[i - 1]]), c2, c4, ax0])
            nc_ = b.node("Slice", [b.node("Shape", [lat[i]]), c0, c2, ax0])
            sizes = b.node("Concat", [nc_, b.node("Cast", [hw_], to=7)], axis=0)
            up = b.node("Resize", [lat[i], roi, b.init(b.uid("noscale"), np.zeros(0, np.float32)), sizes], mode="nearest",
                        coordinate_transformation_mode="asymmetric", nearest_mode="floor")
        else:
            up = b.node("Resize", [lat[i], roi, scales], mode="nearest",
                        coordinate_transformation_mode="asymmetric", nearest_mode="floor")
        lat[i - 1] = b.node("Add", [lat[i - 1], up])
    inter = [conv(lat[i], neck_ch, neck_ch, 3, 1, f"neck.fpn_convs.{i}", False, gain=1.0) for i in range(3)]
    for i in range(2):
        d = conv(inter[i], neck_ch, neck_ch, 3, 2, f"neck.downsample_convs.{i}", False, gain=1.0)
        inter[i + 1] = b.node("Add", [inter[i + 1], d])
    outs = [inter[0]] + [conv(inter[i], neck_ch, neck_ch, 3, 1, f"neck.pafpn_convs.{i - 1}", False, gain=1.0)
                         for i in (1, 2)]

    names = {"score": [], "bbox": [], "kps": []}
    shapes = {"score": 1, "bbox": 4, "kps": 10}
    for i, stride in enumerate((8, 16, 32)):
        h = outs[i]
        c = neck_ch
        for j in range(2):
            h = conv(h, c, c, 3, 1, f"head.{stride}.cls_convs.{j}.dw", True, group=c)
            h = conv(h, c, head_ch, 1, 1, f"head.{stride}.cls_convs.{j}.pw", True)
            c = head_ch
        cls = conv(h, c, 2 * 1, 3, 1, f"head.{stride}.cls", False, gain=cls_gain, bias_mean=cls_bias, bias_std=0.0)
        cls = b.node("Sigmoid", [cls])
        box = conv(h, c, 2 * 4, 3, 1, f"head.{stride}.reg", False, gain=0.5, bias_mean=2.0, bias_std=0.3)
        kps = conv(h, c, 2 * 10, 3, 1, f"head.{stride}.kps", False, gain=0.5, bias_mean=0.0, bias_std=0.5)
        for kind, t in (("score", cls), ("bbox", box), ("kps", kps)):
            t = b.node("Transpose", [t], perm=[0, 2, 3, 1])
            shp = b.init(f"head.{stride}.{kind}.shape", np.array([-1, shapes[kind]], np.int64))
            nm = f"{kind}_{stride}"
            b.node("Reshape", [t, shp], outputs=[nm])
            names[kind].append(nm)
    for kind in ("score", "bbox", "kps"):
        for nm in names[kind]:
            b.add_output(nm, ["A", shapes[kind]])
    return b.save(path)


def make_det_500m(path: str, seed: int = 100, cls_bias: float = -3.0, cls_gain: float = 20.0) -> str:
    return make_scrfd(path, seed=seed, cls_bias=cls_bias, cls_gain=cls_gain)


# ----------------------------------------------------------------------------- pre-decoded detector
def make_predecoded_det(path: str, hw: int = 64, seed: int = 7, three_d: bool = True) -> str:
    """A detector whose only output is already ``[1, N, 15]`` (or ``[N, 15]``) rows
    ``x1,y1,x2,y2,score,10 kps`` — the layout reference postprocess() consumes directly
    (src/face_detector.cpp:242-325).  Three stride-2 convs then NHWC flatten."""
    W = _W(seed)
    b = OnnxBuilder("predecoded")
    x = b.add_input("input.1", [1, 3, hw, hw])
    chans = [3, 8, 16, 15]
    for i in range(3):
        last = i == 2
        w = W.conv(chans[i + 1], chans[i], 3, 2.0)
        if last:
            # columns: x1,y1 around 10..30, x2,y2 around 40..60, score ~ N(.5,.3), kps 20..50
            mean = np.array([20, 20, 50, 50, 0.5] + [35] * 10, np.float32)
            w *= np.array([6, 6, 6, 6, 0.3] + [8] * 10, np.float32)[:, None, None, None]
            bias = mean
        else:
            bias = W.bias(chans[i + 1])
        x = b.node("Conv", [x, b.init(f"c{i}.weight", w), b.init(f"c{i}.bias", bias)],
                   kernel_shape=[3, 3], strides=[2, 2], pads=[1] * 4, dilations=[1, 1], group=1)
        if not last:
            x = b.node("Relu", [x])
    x = b.node("Transpose", [x], perm=[0, 2, 3, 1])
    shp = [1, -1, 15] if three_d else [-1, 15]
    b.node("Reshape", [x, b.init("out.shape", np.array(shp, np.int64))], outputs=["dets"])
    n = (hw // 8) ** 2
    b.add_output("dets", [1, n, 15] if three_d else [n, 15])
    return b.save(path)


def model_cache_dir() -> str:
    d = os.environ.get("FACEHIP_MODEL_DIR", "/tmp/facehip_models")
    os.makedirs(d, exist_ok=True)
    return d


def cached(name: str, maker, *args, **kw) -> str:
    """Build ``name`` into the cache directory once (atomic rename) and return its path."""
    path = os.path.join(model_cache_dir(), name)
    if not os.path.exists(path):
        tmp = f"{path}.tmp{os.getpid()}"
        maker(tmp, *args, **kw)
        os.replace(tmp, path)
    return path
