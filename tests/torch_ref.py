"""Independent evaluation of the ONNX graphs with PyTorch-CPU (fp64 by default).

Used only to pin the oracle; PyTorch is not a reference implementation
and none of this is on the product path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from oracle import onnx_min


def run_graph(path_or_graph, feeds: dict, dtype=torch.float64) -> dict:
    g = onnx_min.load(path_or_graph) if isinstance(path_or_graph, str) else path_or_graph
    env = {}
    for k, v in g.inits.items():
        env[k] = torch.from_numpy(np.asarray(v)).to(dtype) if v.dtype.kind == "f" else torch.from_numpy(np.asarray(v))
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
