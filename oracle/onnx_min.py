"""Minimal ONNX protobuf reader for the ORACLE (test infrastructure only).

Written independently of the product's C++ loader (facerecognizeonnx_amd/csrc/onnx_reader.cpp)
so that a loader bug cannot cancel out in parity tests.  Field numbers: SURVEY.md Appendix C.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np


def _read_varint(buf: memoryview, pos: int):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf: memoryview):
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fno, wt, v


def _s64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v, wt):
    if wt == 0:
        return [_s64(v)]
    out = []
    pos = 0
    while pos < len(v):
        x, pos = _read_varint(v, pos)
        out.append(_s64(x))
    return out


_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _tensor(buf: memoryview):
    dims, dtype, name, raw = [], 1, "", None
    fdata, i64data, i32data = [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v, wt)
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 4:
            fdata += list(struct.unpack(f"<{len(v) // 4}f", bytes(v))) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fno == 7:
            i64data += _packed_varints(v, wt)
        elif fno == 5:
            i32data += _packed_varints(v, wt)
    np_dt = _DT[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_dt).copy()
    elif fdata:
        arr = np.array(fdata, np_dt)
    elif i64data:
        arr = np.array(i64data, np_dt)
    elif i32data:
        arr = np.array(i32data, np_dt)
    else:
        arr = np.zeros(0, np_dt)
    if dims:
        arr = arr.reshape(dims)
    elif arr.size == 1:
        arr = arr.reshape(())
    return name, arr


@dataclass
class Node:
    op: str
    inputs: list
    outputs: list
    attrs: dict = field(default_factory=dict)
    name: str = ""


def _attribute(buf: memoryview):
    name, val = "", None
    floats, ints = [], []
    typ = 0
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _s64(v)
        elif fno == 4:
            val = bytes(v).decode()
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", bytes(v))) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fno == 8:
            ints += _packed_varints(v, wt)
        elif fno == 20:
            typ = v
    if typ == 6 or (val is None and floats):
        val = floats
    elif typ == 7 or (val is None and ints):
        val = ints
    return name, val


def _node(buf: memoryview) -> Node:
    n = Node("", [], [])
    for fno, wt, v in _fields(buf):
        if fno == 1:
            n.inputs.append(bytes(v).decode())
        elif fno == 2:
            n.outputs.append(bytes(v).decode())
        elif fno == 3:
            n.name = bytes(v).decode()
        elif fno == 4:
            n.op = bytes(v).decode()
        elif fno == 5:
            k, val = _attribute(v)
            n.attrs[k] = val
    return n


def _value_info(buf: memoryview):
    name, shape = "", []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            for f2, _, v2 in _fields(v):
                if f2 != 1:
                    continue
                for f3, _, v3 in _fields(v2):
                    if f3 != 2:
                        continue
                    for f4, _, v4 in _fields(v3):
                        if f4 != 1:
                            continue
                        dim = -1
                        for f5, _, v5 in _fields(v4):
                            if f5 == 1:
                                dim = _s64(v5)
                        shape.append(dim)
    return name, shape


@dataclass
class Graph:
    nodes: list
    inits: dict
    inputs: list     # [(name, shape)] excluding initializers
    outputs: list    # [(name, shape)]


def load(path: str) -> Graph:
    with open(path, "rb") as f:
        data = memoryview(f.read())
    gbuf = None
    for fno, wt, v in _fields(data):
        if fno == 7:
            gbuf = v
    if gbuf is None:
        raise ValueError("no graph in model")
    nodes, inits, inputs, outputs = [], {}, [], []
    for fno, wt, v in _fields(gbuf):
        if fno == 1:
            nodes.append(_node(v))
        elif fno == 5:
            k, arr = _tensor(v)
            inits[k] = arr
        elif fno == 11:
            inputs.append(_value_info(v))
        elif fno == 12:
            outputs.append(_value_info(v))
    inputs = [(n, s) for n, s in inputs if n not in inits]
    return Graph(nodes, inits, inputs, outputs)
