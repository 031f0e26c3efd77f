"""Regenerates the golden fixtures in this directory.

  python tests/golden/make_golden.py

* tiny_iresnet.onnx / tiny_scrfd.onnx : seeded synthetic graphs (facerecognizeonnx_amd.synth)
* *_io.npz : seeded inputs and the graph outputs evaluated INDEPENDENTLY of the oracle and of the
  product, with PyTorch-CPU in float64 (oracle/torch_graph.py).  They pin the oracle's graph
  operators; the reference itself has no golden vectors and cannot run here.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from facerecognizeonnx_amd.synth import models  # noqa: E402
from oracle import torch_graph as torch_ref  # noqa: E402


def main():
    r = models.make_iresnet(os.path.join(HERE, "tiny_iresnet.onnx"), (1, 1, 1, 1), (8, 8, 16, 16), 112, 64, seed=11, fold_bn=False)
    rng = np.random.default_rng(5)
    x = ((rng.integers(0, 256, (2, 3, 112, 112)).astype(np.float32) - 127.5) / 128.0).astype(np.float32)
    out = torch_ref.run_graph(r, {"input.1": x})
    np.savez_compressed(os.path.join(HERE, "tiny_iresnet_io.npz"), x=x, y=out["683"].astype(np.float64))

    s = models.make_scrfd(os.path.join(HERE, "tiny_scrfd.onnx"), (1, 1, 1, 1), (8, 8, 8, 16, 16, 24), 8, 16, seed=12, cls_bias=-1.0,
                          static_hw=64)
    x = ((rng.integers(0, 256, (1, 3, 64, 64)).astype(np.float32) - 127.5) / 128.0).astype(np.float32)
    out = torch_ref.run_graph(s, {"input.1": x})
    np.savez_compressed(os.path.join(HERE, "tiny_scrfd_io.npz"), x=x, **{k: v.astype(np.float64) for k, v in out.items()})
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
