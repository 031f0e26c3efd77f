"""ONNX files written by an exporter this build did NOT write: torch.onnx.export (TorchScript exporter, opset 11).

Every other .onnx fixture in this repo comes from `facerecognizeonnx_amd/synth/onnx_writer.py`; the reference accepts any valid graph
(`new Ort::Session(env, path, opts)`, reference src/face_detector.cpp:24-26, src/face_recognizer.cpp:25-27), so the loader / planner
must be exercised on graphs with a foreign exporter's habits: `Identity` nodes that alias de-duplicated initializers, `Constant`
nodes, opset-11 `Resize` with constant roi / scales, `Shape → Gather → …` shape arithmetic under dynamic H / W, Dropout, BatchNorm1d
behind a Gemm, PReLU slopes of shape [C,1,1].

Run in the BUILD container only (`python tests/golden/make_exported_golden.py`); writes

    exported_iresnet_default.onnx   arcface_torch-shaped IResNet with DEFAULT-initialised BN / PReLU parameters — the exporter
                                    de-duplicates the equal tensors into Identity(initializer) aliases
    exported_iresnet_trained.onnx   the same module with randomised ("trained-looking") parameters and running statistics
    exported_scrfd.onnx             SCRFD-shaped detector: depthwise-separable backbone, F.interpolate x2 + add, per-stride heads,
                                    sigmoid, permute + reshape, dynamic H / W
    exported_scrfd_static.onnx      the same module exported with a fixed 1 x 3 x 96 x 128 input (no dynamic axes)
    exported_io.npz                 seeded inputs + the modules' own torch float64 outputs

torch is a third-party evaluator here, not the reference; nothing of /root/reference is read.  The `onnx` python package is absent
in this image: the exporter's only use of it (`onnx_proto_utils._add_onnxscript_fn`, a post-pass for onnx-script functions, of which
these graphs have none) is bypassed.
"""
from __future__ import annotations

import io
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ arcface_torch `iresnet` shape (public architecture, restated)
class IBasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(inplanes, eps=1e-5)
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, eps=1e-5)
        self.prelu = nn.PReLU(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes, eps=1e-5)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.bn1(x)
        out = self.conv1(out)
        out = self.bn2(out)
        out = self.prelu(out)
        out = self.conv2(out)
        out = self.bn3(out)
        if self.downsample is not None:
            identity = self.downsample(x)
        return out + identity


class IResNet(nn.Module):
    def __init__(self, layers=(1, 2, 2, 1), widths=(16, 32, 64, 128), size=112, feat=32, dropout=0.4):
        super().__init__()
        self.inplanes = widths[0]
        self.conv1 = nn.Conv2d(3, widths[0], 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(widths[0], eps=1e-5)
        self.prelu = nn.PReLU(widths[0])
        self.layer1 = self._make(widths[0], layers[0])
        self.layer2 = self._make(widths[1], layers[1])
        self.layer3 = self._make(widths[2], layers[2])
        self.layer4 = self._make(widths[3], layers[3])
        self.bn2 = nn.BatchNorm2d(widths[3], eps=1e-5)
        self.dropout = nn.Dropout(p=dropout, inplace=True)
        fc_scale = (size // 16) ** 2
        self.fc = nn.Linear(widths[3] * fc_scale, feat)
        self.features = nn.BatchNorm1d(feat, eps=1e-5)

    def _make(self, planes, blocks):
        down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, 2, bias=False), nn.BatchNorm2d(planes, eps=1e-5))
        seq = [IBasicBlock(self.inplanes, planes, 2, down)]
        self.inplanes = planes
        for _ in range(1, blocks):
            seq.append(IBasicBlock(planes, planes))
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.prelu(self.bn1(self.conv1(x)))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.bn2(x)
        x = torch.flatten(x, 1)
        x = self.dropout(x)
        x = self.fc(x)
        return self.features(x)


# ------------------------------------------------------------------ SCRFD-shaped detector (mmdet MobileNetV1 + PAFPN + SCRFDHead shape)
def _dwsep(cin, cout, stride):
    return nn.Sequential(nn.Conv2d(cin, cin, 3, stride, 1, groups=cin, bias=False), nn.BatchNorm2d(cin), nn.ReLU(inplace=True),
                         nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class ScrfdLike(nn.Module):
    def __init__(self, widths=(8, 16, 24, 32, 48), fpn=16, head=32, num_anchors=2):
        super().__init__()
        w = widths
        self.stem = nn.Sequential(nn.Conv2d(3, w[0], 3, 2, 1, bias=False), nn.BatchNorm2d(w[0]), nn.ReLU(inplace=True),
                                  _dwsep(w[0], w[1], 1))
        self.s4 = nn.Sequential(_dwsep(w[1], w[1], 2), _dwsep(w[1], w[1], 1))
        self.s8 = nn.Sequential(_dwsep(w[1], w[2], 2), _dwsep(w[2], w[2], 1))
        self.s16 = nn.Sequential(_dwsep(w[2], w[3], 2), _dwsep(w[3], w[3], 1))
        self.s32 = nn.Sequential(_dwsep(w[3], w[4], 2), _dwsep(w[4], w[4], 1))
        self.lat = nn.ModuleList([nn.Conv2d(c, fpn, 1) for c in (w[2], w[3], w[4])])
        self.fpn_out = nn.ModuleList([nn.Conv2d(fpn, fpn, 3, 1, 1) for _ in range(3)])
        self.down = nn.ModuleList([nn.Conv2d(fpn, fpn, 3, 2, 1) for _ in range(2)])
        self.pa_out = nn.ModuleList([nn.Conv2d(fpn, fpn, 3, 1, 1) for _ in range(2)])
        self.tower = nn.ModuleList([nn.Sequential(_dwsep(fpn, head, 1), _dwsep(head, head, 1)) for _ in range(3)])
        self.cls = nn.ModuleList([nn.Conv2d(head, num_anchors, 3, 1, 1) for _ in range(3)])
        self.reg = nn.ModuleList([nn.Conv2d(head, 4 * num_anchors, 3, 1, 1) for _ in range(3)])
        self.kps = nn.ModuleList([nn.Conv2d(head, 10 * num_anchors, 3, 1, 1) for _ in range(3)])
        self.scales = nn.ParameterList([nn.Parameter(torch.tensor(1.0)) for _ in range(3)])

    def forward(self, x):
        c2 = self.s4(self.stem(x))
        c3 = self.s8(c2)
        c4 = self.s16(c3)
        c5 = self.s32(c4)
        l3, l4, l5 = self.lat[0](c3), self.lat[1](c4), self.lat[2](c5)
        l4 = l4 + F.interpolate(l5, scale_factor=2.0, mode="nearest")
        l3 = l3 + F.interpolate(l4, scale_factor=2.0, mode="nearest")
        p3, p4, p5 = self.fpn_out[0](l3), self.fpn_out[1](l4), self.fpn_out[2](l5)
        p4 = self.pa_out[0](p4 + self.down[0](p3))
        p5 = self.pa_out[1](p5 + self.down[1](p4))
        cls, reg, kps = [], [], []
        for i, p in enumerate((p3, p4, p5)):
            t = self.tower[i](p)
            cls.append(torch.sigmoid(self.cls[i](t)).permute(0, 2, 3, 1).reshape(-1, 1))
            reg.append((self.reg[i](t) * self.scales[i]).permute(0, 2, 3, 1).reshape(-1, 4))
            kps.append(self.kps[i](t).permute(0, 2, 3, 1).reshape(-1, 10))
        return tuple(cls + reg + kps)


def _randomise(m: nn.Module, seed: int):
    """'Trained-looking' parameters: SURVEY.md §8(d)'s distributions (BN gamma ~ U(.5,1.5), beta / mean ~ N(0,.1), var ~ U(.5,1.5),
    PReLU slope .25 * U(.5,1.5), conv ~ N(0, 2 / fan_in))."""
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
            c = mod.num_features
            mod.weight.data = torch.rand(c, generator=g) + 0.5
            mod.bias.data = torch.randn(c, generator=g) * 0.1
            mod.running_mean = torch.randn(c, generator=g) * 0.1
            mod.running_var = torch.rand(c, generator=g) + 0.5
        elif isinstance(mod, nn.PReLU):
            mod.weight.data = 0.25 * (torch.rand(mod.num_parameters, generator=g) + 0.5)
        elif isinstance(mod, nn.Conv2d):
            fan = mod.in_channels // mod.groups * mod.kernel_size[0] ** 2
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * (2.0 / fan) ** 0.5
            if mod.bias is not None:
                mod.bias.data = torch.randn(mod.bias.shape, generator=g) * 0.05
        elif isinstance(mod, nn.Linear):
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * (1.0 / mod.in_features) ** 0.5
            mod.bias.data = torch.randn(mod.bias.shape, generator=g) * 0.05
    if isinstance(m, ScrfdLike):
        for k, s in enumerate(m.scales):
            s.data = torch.tensor(0.8 + 0.3 * k)


def _export(module: nn.Module, example: torch.Tensor, path: str, input_name: str, output_names, dynamic_axes):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    orig = onnx_proto_utils._add_onnxscript_fn
    onnx_proto_utils._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes     # needs the absent `onnx` package; no-op here
    try:
        buf = io.BytesIO()
        torch.onnx.export(module, (example,), buf, dynamo=False, opset_version=11, input_names=[input_name],
                          output_names=list(output_names), dynamic_axes=dynamic_axes, do_constant_folding=True)
    finally:
        onnx_proto_utils._add_onnxscript_fn = orig
    with open(path, "wb") as f:
        f.write(buf.getvalue())
    return len(buf.getvalue())


def main():
    torch.manual_seed(0)
    rng = np.random.default_rng(77)
    out = {}

    # IResNet, default-initialised BN / PReLU (equal tensors -> Identity aliases) and trained-looking
    for tag, seed in (("default", None), ("trained", 11)):
        net = IResNet().eval()
        if seed is None:
            g = torch.Generator().manual_seed(5)        # convs / fc random (He), BN + PReLU left at their defaults
            for mod in net.modules():
                if isinstance(mod, nn.Conv2d):
                    mod.weight.data = torch.randn(mod.weight.shape, generator=g) * (2.0 / (mod.in_channels * mod.kernel_size[0] ** 2)) ** 0.5
        else:
            _randomise(net, seed)
        x = ((rng.integers(0, 256, (3, 3, 112, 112)).astype(np.float32)) - 127.5) / 128.0
        path = os.path.join(HERE, f"exported_iresnet_{tag}.onnx")
        n = _export(net, torch.from_numpy(x), path, "input.1", ["embedding"], {"input.1": {0: "N"}, "embedding": {0: "N"}})
        with torch.no_grad():
            y = net.double()(torch.from_numpy(x).double()).numpy()
        out[f"iresnet_{tag}_x"] = x
        out[f"iresnet_{tag}_y"] = y
        print(f"exported_iresnet_{tag}.onnx: {n} bytes, output {y.shape}, |y| max {np.abs(y).max():.3f}")

    det = ScrfdLike().eval()
    _randomise(det, 23)
    # a bias on the classification branches so that some scores pass 0.5
    for c in det.cls:
        c.bias.data += 0.2
    xd = ((rng.integers(0, 256, (1, 3, 96, 128)).astype(np.float32)) - 127.5) / 128.0
    names = [f"score_{s}" for s in (8, 16, 32)] + [f"bbox_{s}" for s in (8, 16, 32)] + [f"kps_{s}" for s in (8, 16, 32)]
    path = os.path.join(HERE, "exported_scrfd.onnx")
    n = _export(det, torch.from_numpy(xd), path, "input.1", names, {"input.1": {0: "N", 2: "H", 3: "W"}})
    out["scrfd_x"] = xd
    with torch.no_grad():
        ys = det.double()(torch.from_numpy(xd).double())
    for nm, y in zip(names, ys):
        out["scrfd_" + nm] = y.numpy()
    # the same graph at a second input size: the dynamic-axes shape arithmetic must fold for any H / W
    xd2 = ((rng.integers(0, 256, (1, 3, 160, 96)).astype(np.float32)) - 127.5) / 128.0
    out["scrfd2_x"] = xd2
    with torch.no_grad():
        ys = det(torch.from_numpy(xd2).double())
    for nm, y in zip(names, ys):
        out["scrfd2_" + nm] = y.numpy()
    print(f"exported_scrfd.onnx: {n} bytes, outputs {[tuple(out['scrfd_' + k].shape) for k in names]}")
    # static-shape export of the same module (what `loadModel` adopts as its input size, reference src/face_detector.cpp:39-57): the
    # shape arithmetic is folded by the exporter here, the Reshape targets / Resize scales are plain constants
    n = _export(det.float(), torch.from_numpy(xd), os.path.join(HERE, "exported_scrfd_static.onnx"), "input.1", names, None)
    print(f"exported_scrfd_static.onnx: {n} bytes (input fixed at 96 x 128)")
    np.savez_compressed(os.path.join(HERE, "exported_io.npz"), **out)


if __name__ == "__main__":
    main()
