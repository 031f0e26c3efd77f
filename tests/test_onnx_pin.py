"""Independent pin of the C++ ONNX reader (csrc/onnx_reader.cpp): the same .onnx files decoded by google.protobuf from a
hand-declared descriptor of the public onnx.proto subset (field numbers from the ONNX IR spec), rendered with the same
canonical text as `fh_onnx_dump`, must give byte-identical text.

Why: reader (C++), writer (synth/onnx_writer.py) and the oracle's loader (oracle/onnx_min.py) share one author and one reading of
the wire format; google.protobuf is a third-party decoder that shares nothing with them.  The `onnx` package itself is not in
this image, so the message descriptors are declared here by hand.  This does not turn parity green (the genuine det_500m.onnx /
w600k_r50.onnx are not available offline): it narrows the risk that the reader mis-parses a conforming file.
CPU only: no GPU call.
"""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import facerecognizeonnx_amd as fa
from facerecognizeonnx_amd.synth import models
from tests import util

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

T = descriptor_pb2.FieldDescriptorProto


def _build_messages():
    f = descriptor_pb2.FileDescriptorProto(name="onnx_subset.proto", package="onnxpin", syntax="proto2")

    def msg(name, fields, nested=()):
        m = f.message_type.add(name=name) if isinstance(name, str) else name
        for (fname, num, ftype, label, tname, packed) in fields:
            fd = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                fd.type_name = tname
            if packed:
                fd.options.packed = True
        return m

    O, R = T.LABEL_OPTIONAL, T.LABEL_REPEATED
    msg("TensorProto", [("dims", 1, T.TYPE_INT64, R, "", False), ("data_type", 2, T.TYPE_INT32, O, "", False),
                        ("float_data", 4, T.TYPE_FLOAT, R, "", True), ("int32_data", 5, T.TYPE_INT32, R, "", True),
                        ("int64_data", 7, T.TYPE_INT64, R, "", True), ("name", 8, T.TYPE_STRING, O, "", False),
                        ("raw_data", 9, T.TYPE_BYTES, O, "", False), ("double_data", 10, T.TYPE_DOUBLE, R, "", True)])
    msg("AttributeProto", [("name", 1, T.TYPE_STRING, O, "", False), ("f", 2, T.TYPE_FLOAT, O, "", False), ("i", 3, T.TYPE_INT64, O, "", False),
                           ("s", 4, T.TYPE_BYTES, O, "", False), ("t", 5, T.TYPE_MESSAGE, O, ".onnxpin.TensorProto", False),
                           ("floats", 7, T.TYPE_FLOAT, R, "", False), ("ints", 8, T.TYPE_INT64, R, "", False), ("type", 20, T.TYPE_INT32, O, "", False)])
    msg("NodeProto", [("input", 1, T.TYPE_STRING, R, "", False), ("output", 2, T.TYPE_STRING, R, "", False), ("name", 3, T.TYPE_STRING, O, "", False),
                      ("op_type", 4, T.TYPE_STRING, O, "", False), ("attribute", 5, T.TYPE_MESSAGE, R, ".onnxpin.AttributeProto", False)])
    msg("Dimension", [("dim_value", 1, T.TYPE_INT64, O, "", False), ("dim_param", 2, T.TYPE_STRING, O, "", False)])
    msg("TensorShapeProto", [("dim", 1, T.TYPE_MESSAGE, R, ".onnxpin.Dimension", False)])
    msg("TypeTensor", [("elem_type", 1, T.TYPE_INT32, O, "", False), ("shape", 2, T.TYPE_MESSAGE, O, ".onnxpin.TensorShapeProto", False)])
    msg("TypeProto", [("tensor_type", 1, T.TYPE_MESSAGE, O, ".onnxpin.TypeTensor", False)])
    msg("ValueInfoProto", [("name", 1, T.TYPE_STRING, O, "", False), ("type", 2, T.TYPE_MESSAGE, O, ".onnxpin.TypeProto", False)])
    msg("GraphProto", [("node", 1, T.TYPE_MESSAGE, R, ".onnxpin.NodeProto", False), ("name", 2, T.TYPE_STRING, O, "", False),
                       ("initializer", 5, T.TYPE_MESSAGE, R, ".onnxpin.TensorProto", False),
                       ("input", 11, T.TYPE_MESSAGE, R, ".onnxpin.ValueInfoProto", False),
                       ("output", 12, T.TYPE_MESSAGE, R, ".onnxpin.ValueInfoProto", False)])
    msg("ModelProto", [("ir_version", 1, T.TYPE_INT64, O, "", False), ("graph", 7, T.TYPE_MESSAGE, O, ".onnxpin.GraphProto", False)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:                                               # older protobuf
        fac = message_factory.MessageFactory(pool)
        return fac.GetPrototype(pool.FindMessageTypeByName("onnxpin.ModelProto"))
    return get(pool.FindMessageTypeByName("onnxpin.ModelProto"))


def _g9(x):
    return "%.9g" % x


def _tensor_values(t):
    """(values as the reader stores them: float32 list or int list, is_float)"""
    if t.raw_data:
        raw = bytes(t.raw_data)
        if t.data_type == 1:
            return list(struct.unpack("<%df" % (len(raw) // 4), raw)), True
        if t.data_type == 7:
            return list(struct.unpack("<%dq" % (len(raw) // 8), raw)), False
        if t.data_type == 6:
            return list(struct.unpack("<%di" % (len(raw) // 4), raw)), False
        if t.data_type == 11:
            return [float(np.float32(v)) for v in struct.unpack("<%dd" % (len(raw) // 8), raw)], True
        raise AssertionError(f"dtype {t.data_type}")
    if t.float_data:
        return [float(np.float32(v)) for v in t.float_data], True
    if t.double_data:
        return [float(np.float32(v)) for v in t.double_data], True
    if t.int64_data:
        return list(t.int64_data), False
    return list(t.int32_data), False


def protobuf_dump(path):
    Model = _build_messages()
    m = Model()
    m.ParseFromString(open(path, "rb").read())
    g = m.graph
    inits = {t.name: t for t in g.initializer}
    dims = lambda d: ",".join(str(int(x)) for x in d)
    shape = lambda v: dims([(dm.dim_value if dm.HasField("dim_value") else -1) for dm in v.type.tensor_type.shape.dim])
    out = []
    for v in g.input:
        if v.name not in inits:
            out.append(f"input {v.name} [{shape(v)}]")
    for v in g.output:
        out.append(f"output {v.name} [{shape(v)}]")
    for name in sorted(inits, key=lambda s: s.encode()):
        t = inits[name]
        vals, is_f = _tensor_values(t)
        n = len(vals)
        if n:
            s = 0.0                                              # sequential fp64 sum of the stored values, as the reader's dump does
            for x in vals:
                s += float(x)
            first, last = float(vals[0]), float(vals[-1])
        else:
            s = first = last = 0.0
        out.append(f"init {name} dtype={t.data_type} [{dims(t.dims)}] n={n} sum={_g9(s)} first={_g9(first)} last={_g9(last)}")
    for nd in g.node:
        line = f"node {nd.op_type} in={','.join(nd.input)} out={','.join(nd.output)}"
        for a in sorted(nd.attribute, key=lambda a: a.name.encode()):
            if len(a.ints):
                val = "ints:" + dims(a.ints)
            elif len(a.floats):
                val = "floats:" + ",".join(_g9(float(np.float32(x))) for x in a.floats)
            elif a.s:
                val = "s:" + a.s.decode()
            elif a.HasField("t") and (len(a.t.dims) or a.t.raw_data or len(a.t.float_data) or len(a.t.int64_data)):
                val = f"t:[{dims(a.t.dims)}]"
            elif a.f != 0.0:
                val = "f:" + _g9(float(np.float32(a.f)))
            else:
                val = "i:" + str(int(a.i))
            line += f" {a.name}={val}"
        out.append(line)
    return "\n".join(out) + "\n"


def reader_dump(path):
    buf = C.create_string_buffer(1 << 24)
    n = fa.lib().fh_onnx_dump(str(path).encode(), buf, len(buf))
    assert n >= 0, fa._lib.last_error()
    return buf.value.decode()


def _files(models_dir):
    yield os.path.join(util.GOLDEN, "tiny_scrfd.onnx")
    yield os.path.join(util.GOLDEN, "tiny_iresnet.onnx")
    yield util.tiny_mbf(models_dir, fold_bn=False)
    yield util.tiny_iresnet(models_dir, fold_bn=False)
    yield util.tiny_scrfd(models_dir, hw=None)                      # dynamic H / W: shape sub-graph nodes (Shape, Gather, Concat, ...)
    yield models.make_predecoded_det(os.path.join(models_dir, "pin_pre.onnx"), 64, 7, True)


def test_reader_agrees_with_google_protobuf(models_dir):
    n_nodes = 0
    for path in _files(models_dir):
        want, got = protobuf_dump(path), reader_dump(path)
        assert got == want, (path, next((a, b) for a, b in zip(got.splitlines(), want.splitlines()) if a != b))
        n_nodes += sum(l.startswith("node ") for l in got.splitlines())
    assert n_nodes > 150


def test_full_size_models_agree_with_google_protobuf():
    """The full-size synthetic det_500m (dynamic axes, Resize, merged heads) and w600k_r50 (174 MB of raw_data) through both decoders."""
    for name, maker in (("det_500m_seed100.onnx", models.make_det_500m), ("w600k_r50_seed200.onnx", models.make_w600k_r50)):
        path = models.cached(name, maker)
        assert reader_dump(path) == protobuf_dump(path), name
