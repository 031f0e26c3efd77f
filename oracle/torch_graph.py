"""Independent evaluation of the ONNX graphs with PyTorch-CPU (test infrastructure, like everything under oracle/).

Two users: (1) the tests pin the oracle's graph evaluation against it in fp64; (2) bench.py's second CPU leg
(`cpu_baseline_torch`) times the two graphs in fp32 on 4 threads as a PROXY for what an optimised CPU engine such as the ONNX
Runtime CPU EP (the reference's engine, src/face_detector.cpp:10-11) delivers — oneDNN convolutions instead of the oracle's
naive loops.  PyTorch is not a reference implementation and none of this is on the product path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle import onnx_min


class TorchGraph:
    """One loaded graph with its initializers converted once (the 174 MB of w600k_r50 are not re-wrapped per call)."""

    def __init__(self, path_or_graph, dtype=torch.float64):
        self.g = onnx_min.load(path_or_graph) if isinstance(path_or_graph, str) else path_or_graph
        self.dtype = dtype
        self.inits = {k: (torch.from_numpy(np.asarray(v)).to(dtype) if v.dtype.kind == "f" else torch.from_numpy(np.asarray(v)))
                      for k, v in self.g.inits.items()}

    def run(self, feeds: dict) -> dict:
        with torch.no_grad():
            return _run(self.g, self.inits, feeds, self.dtype)


def run_graph(path_or_graph, feeds: dict, dtype=torch.float64) -> dict:
    return TorchGraph(path_or_graph, dtype).run(feeds)


def _run(g, inits, feeds, dtype) -> dict:
    env = dict(inits)
    for k, v in feeds.items():
        env[k] = torch.from_numpy(np.asarray(v)).to(dtype)
    for n in g.nodes:
        a = n.attrs
        i = [env[k] if k else None for k in n.inputs]
        if n.op == "Conv":
            y = F.conv2d(i[0], i[1], i[2] if len(i) > 2 else None, stride=a.get("strides", [1, 1]),
                         padding=a.get("pads", [0, 0, 0, 0])[:2], groups=a.get("group", 1))
        elif n.op == "BatchNormalization":
            y = F.batch_norm(i[0], i[3], i[4], i[1], i[2], False, 0.0, a.get("epsilon", 1e-5))
        elif n.op == "PRelu":
            y = F.prelu(i[0], i[1].reshape(-1))
        elif n.op == "Relu":
            y = F.relu(i[0])
        elif n.op == "Sigmoid":
            y = torch.sigmoid(i[0])
        elif n.op == "Add":
            y = i[0] + i[1]
        elif n.op in ("Identity", "Dropout"):                   # inference: pass-through
            y = i[0]
        elif n.op in ("Mul", "Sub", "Div"):                     # element-wise, numpy-style broadcasting (constant operands)
            y = {"Mul": torch.mul, "Sub": torch.sub, "Div": torch.div}[n.op](i[0], i[1])
        elif n.op == "Shape":
            y = torch.tensor(list(i[0].shape), dtype=torch.int64)
        elif n.op == "Slice":
            y = i[0][int(i[1][0]):int(i[2][0])]
        elif n.op == "Concat":
            y = torch.cat([v.reshape(-1) for v in i])
        elif n.op == "Cast":
            y = i[0].to(torch.int64)
        elif n.op == "Resize":
            s = int(i[2][2]) if len(i) > 2 and i[2] is not None and i[2].numel() else int(i[3][2]) // i[0].shape[2]
            y = F.interpolate(i[0], scale_factor=s, mode="nearest")
        elif n.op == "Transpose":
            y = i[0].permute(*a["perm"]).contiguous()
        elif n.op == "Reshape":
            y = i[0].reshape([int(d) for d in i[1]])
        elif n.op == "Flatten":
            y = i[0].flatten(1)
        elif n.op == "MatMul":
            y = i[0] @ i[1]
        elif n.op == "Gemm":
            y = F.linear(i[0], i[1], i[2] if len(i) > 2 else None)
        else:
            raise NotImplementedError(n.op)
        env[n.outputs[0]] = y
    return {name: env[name].numpy() for name, _ in g.outputs}
