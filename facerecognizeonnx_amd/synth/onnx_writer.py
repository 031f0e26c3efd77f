"""Dependency-free ONNX (protobuf wire format) writer.

Only the message fields the two face graphs need:
ModelProto{ir_version=1, producer_name=2, graph=7, opset_import=8},
GraphProto{node=1, name=2, initializer=5, input=11, output=12},
NodeProto{input=1, output=2, name=3, op_type=4, attribute=5},
AttributeProto{name=1, f=2, i=3, s=4, t=5, floats=7, ints=8, type=20},
TensorProto{dims=1, data_type=2, name=8, raw_data=9},
ValueInfoProto{name=1, type=2}.
The `onnx` python package is not installed in this image, hence the hand encoder.
"""
from __future__ import annotations

import struct
from typing import Iterable, Sequence

import numpy as np

FLOAT, INT64 = 1, 7
_ATTR_FLOAT, _ATTR_INT, _ATTR_STRING, _ATTR_TENSOR, _ATTR_FLOATS, _ATTR_INTS = 1, 2, 3, 4, 6, 7


def _varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _ld(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def _vi(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(v)


def _str(field: int, s: str) -> bytes:
    return _ld(field, s.encode("utf-8"))


def tensor_proto(name: str, arr: np.ndarray) -> bytes:
    if arr.dtype == np.float32:
        dt = FLOAT
    elif arr.dtype == np.int64:
        dt = INT64
    else:
        raise TypeError(arr.dtype)
    out = bytearray()
    for d in arr.shape:
        out += _vi(1, int(d))
    out += _vi(2, dt)
    out += _str(8, name)
    out += _ld(9, np.ascontiguousarray(arr).tobytes())
    return bytes(out)


def _attr(name: str, value) -> bytes:
    out = bytearray(_str(1, name))
    if isinstance(value, float):
        out += _key(2, 5) + struct.pack("<f", value)
        out += _vi(20, _ATTR_FLOAT)
    elif isinstance(value, (int, np.integer)):
        out += _vi(3, int(value))
        out += _vi(20, _ATTR_INT)
    elif isinstance(value, str):
        out += _ld(4, value.encode())
        out += _vi(20, _ATTR_STRING)
    elif isinstance(value, np.ndarray):
        out += _ld(5, tensor_proto("", value))
        out += _vi(20, _ATTR_TENSOR)
    elif isinstance(value, (list, tuple)) and value and isinstance(value[0], float):
        out += _ld(7, b"".join(struct.pack("<f", v) for v in value))  # packed
        out += _vi(20, _ATTR_FLOATS)
    elif isinstance(value, (list, tuple)):
        out += _ld(8, b"".join(_varint(int(v)) for v in value))  # packed
        out += _vi(20, _ATTR_INTS)
    else:
        raise TypeError(type(value))
    return bytes(out)


def _value_info(name: str, shape: Sequence) -> bytes:
    dims = bytearray()
    for d in shape:
        if isinstance(d, str):
            dims += _ld(1, _str(2, d))       # dim_param  → dynamic (-1 in ORT)
        else:
            dims += _ld(1, _vi(1, int(d)))   # dim_value
    tensor_type = _vi(1, FLOAT) + _ld(2, bytes(dims))
    return _str(1, name) + _ld(2, _ld(1, tensor_type))


class OnnxBuilder:
    """Accumulates nodes / initializers and serialises a ModelProto."""

    def __init__(self, graph_name: str = "g", opset: int = 11):
        self.nodes: list[bytes] = []
        self.inits: list[bytes] = []
        self.inputs: list[bytes] = []
        self.outputs: list[bytes] = []
        self.graph_name = graph_name
        self.opset = opset
        self._uid = 0

    def uid(self, prefix: str) -> str:
        self._uid += 1
        return f"{prefix}_{self._uid}"

    def add_input(self, name: str, shape: Sequence) -> str:
        self.inputs.append(_value_info(name, shape))
        return name

    def add_output(self, name: str, shape: Sequence) -> str:
        self.outputs.append(_value_info(name, shape))
        return name

    def init(self, name: str, arr: np.ndarray) -> str:
        self.inits.append(tensor_proto(name, arr))
        return name

    def node(self, op: str, inputs: Iterable[str], outputs: Iterable[str] | None = None, **attrs) -> str:
        outs = list(outputs) if outputs else [self.uid(op.lower())]
        b = bytearray()
        for i in inputs:
            b += _str(1, i)
        for o in outs:
            b += _str(2, o)
        b += _str(3, self.uid("n"))
        b += _str(4, op)
        for k, v in attrs.items():
            b += _ld(5, _attr(k, v))
        self.nodes.append(bytes(b))
        return outs[0]

    def serialize(self) -> bytes:
        g = bytearray()
        for n in self.nodes:
            g += _ld(1, n)
        g += _str(2, self.graph_name)
        for t in self.inits:
            g += _ld(5, t)
        for i in self.inputs:
            g += _ld(11, i)
        for o in self.outputs:
            g += _ld(12, o)
        m = bytearray()
        m += _vi(1, 6)                                  # ir_version
        m += _str(2, "facehip-synth")                   # producer_name
        m += _ld(7, bytes(g))
        m += _ld(8, _str(1, "") + _vi(2, self.opset))   # opset_import
        return bytes(m)

    def save(self, path: str) -> str:
        with open(path, "wb") as f:
            f.write(self.serialize())
        return path
