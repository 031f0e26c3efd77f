// onnx_reader.cpp — protobuf wire-format walk over ModelProto → GraphProto (field numbers:
// SURVEY.md Appendix C).  No protobuf library, no onnx schema package.
#include "onnx_reader.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace fh {
namespace {

struct Span {
    const uint8_t* p;
    const uint8_t* e;
    bool done() const { return p >= e; }
};

uint64_t varint(Span& s) {
    uint64_t v = 0;
    int shift = 0;
    while (true) {
        if (s.p >= s.e || shift > 63) throw std::runtime_error("onnx: truncated varint");
        uint8_t b = *s.p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return v;
        shift += 7;
    }
}

struct Field {
    int no;
    int wt;
    uint64_t v = 0;   // wt 0 / 1 / 5 payload
    Span sub{nullptr, nullptr};
};

bool next(Span& s, Field& f) {
    if (s.done()) return false;
    uint64_t key = varint(s);
    f.no = (int)(key >> 3);
    f.wt = (int)(key & 7);
    switch (f.wt) {
        case 0: f.v = varint(s); break;
        case 1:
            if (s.e - s.p < 8) throw std::runtime_error("onnx: truncated fixed64");
            memcpy(&f.v, s.p, 8); s.p += 8; break;
        case 2: {
            uint64_t n = varint(s);
            if ((uint64_t)(s.e - s.p) < n) throw std::runtime_error("onnx: truncated field");
            f.sub = Span{s.p, s.p + n};
            s.p += n;
            break;
        }
        case 5: {
            if (s.e - s.p < 4) throw std::runtime_error("onnx: truncated fixed32");
            uint32_t u; memcpy(&u, s.p, 4); s.p += 4; f.v = u; break;
        }
        default: throw std::runtime_error("onnx: unsupported wire type");
    }
    return true;
}

std::string str(const Span& s) { return std::string((const char*)s.p, (size_t)(s.e - s.p)); }

float f32_of(uint64_t v) { uint32_t u = (uint32_t)v; float f; memcpy(&f, &u, 4); return f; }

void read_ints(const Field& f, std::vector<int64_t>& out) {
    if (f.wt == 0) { out.push_back((int64_t)f.v); return; }
    Span s = f.sub;
    while (!s.done()) out.push_back((int64_t)varint(s));
}

void read_floats(const Field& f, std::vector<float>& out) {
    if (f.wt == 5) { out.push_back(f32_of(f.v)); return; }
    size_t n = (size_t)(f.sub.e - f.sub.p) / 4;
    size_t base = out.size();
    out.resize(base + n);
    if (n) memcpy(out.data() + base, f.sub.p, n * 4);
}

OnnxTensor read_tensor(Span s) {
    OnnxTensor t;
    Span raw{nullptr, nullptr};
    std::vector<double> dd;
    Field f;
    while (next(s, f)) {
        switch (f.no) {
            case 1: read_ints(f, t.dims); break;
            case 2: t.dtype = (int)f.v; break;
            case 4: read_floats(f, t.f); break;
            case 5: read_ints(f, t.i); break;       // int32_data
            case 7: read_ints(f, t.i); break;       // int64_data
            case 8: t.name = str(f.sub); break;
            case 9: raw = f.sub; break;
            case 10:
                if (f.wt == 1) { double d; memcpy(&d, &f.v, 8); dd.push_back(d); }
                else { size_t n = (size_t)(f.sub.e - f.sub.p) / 8; size_t b = dd.size(); dd.resize(b + n); if (n) memcpy(dd.data() + b, f.sub.p, n * 8); }
                break;
            case 13: case 14:
                if (f.no == 14 && f.v == 1) throw std::runtime_error("onnx: external tensor data not supported");
                break;
            default: break;
        }
    }
    size_t nbytes = raw.p ? (size_t)(raw.e - raw.p) : 0;
    if (raw.p) {
        switch (t.dtype) {
            case 1: t.f.resize(nbytes / 4); if (nbytes >= 4) memcpy(t.f.data(), raw.p, nbytes / 4 * 4); break;      // (an empty payload: data() may be null)
            case 7: t.i.resize(nbytes / 8); if (nbytes >= 8) memcpy(t.i.data(), raw.p, nbytes / 8 * 8); break;
            case 6: { t.i.resize(nbytes / 4); for (size_t k = 0; k < nbytes / 4; ++k) { int32_t v; memcpy(&v, raw.p + 4 * k, 4); t.i[k] = v; } break; }
            case 11: { t.f.resize(nbytes / 8); for (size_t k = 0; k < nbytes / 8; ++k) { double v; memcpy(&v, raw.p + 8 * k, 8); t.f[k] = (float)v; } break; }
            default: throw std::runtime_error("onnx: unsupported tensor data_type " + std::to_string(t.dtype));
        }
    } else if (!dd.empty()) {
        t.f.assign(dd.begin(), dd.end());
    }
    return t;
}

OnnxNode read_node(Span s) {
    OnnxNode n;
    Field f;
    while (next(s, f)) {
        switch (f.no) {
            case 1: n.inputs.push_back(str(f.sub)); break;
            case 2: n.outputs.push_back(str(f.sub)); break;
            case 3: n.name = str(f.sub); break;
            case 4: n.op = str(f.sub); break;
            case 5: {
                Span a = f.sub;
                Field g;
                std::string name;
                OnnxAttr at;
                while (next(a, g)) {
                    switch (g.no) {
                        case 1: name = str(g.sub); break;
                        case 2: at.f = f32_of(g.v); break;
                        case 3: at.i = (int64_t)g.v; break;
                        case 4: at.s = str(g.sub); break;
                        case 5: at.t = read_tensor(g.sub); break;
                        case 7: read_floats(g, at.floats); break;
                        case 8: read_ints(g, at.ints); break;
                        default: break;
                    }
                }
                n.attrs[name] = std::move(at);
                break;
            }
            default: break;
        }
    }
    return n;
}

OnnxValueInfo read_value_info(Span s) {
    OnnxValueInfo vi;
    Field f;
    while (next(s, f)) {
        if (f.no == 1) vi.name = str(f.sub);
        else if (f.no == 2) {                                   // TypeProto
            Span ty = f.sub; Field a;
            while (next(ty, a)) {
                if (a.no != 1) continue;                        // tensor_type
                Span tt = a.sub; Field b;
                while (next(tt, b)) {
                    if (b.no != 2) continue;                    // shape
                    Span sh = b.sub; Field c;
                    while (next(sh, c)) {
                        if (c.no != 1) continue;                // dim
                        Span dm = c.sub; Field d;
                        int64_t val = -1;                       // dim_param / unset → dynamic
                        while (next(dm, d)) if (d.no == 1) val = (int64_t)d.v;
                        vi.shape.push_back(val);
                    }
                }
            }
        }
    }
    return vi;
}

}  // namespace

OnnxModel load_onnx(const std::string& path) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("onnx: cannot open " + path);
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)(sz > 0 ? sz : 0));
    size_t got = buf.empty() ? 0 : fread(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    if (got != buf.size() || buf.empty()) throw std::runtime_error("onnx: cannot read " + path);

    Span top{buf.data(), buf.data() + buf.size()};
    Span graph{nullptr, nullptr};
    Field f;
    while (next(top, f)) if (f.no == 7 && f.wt == 2) graph = f.sub;
    if (!graph.p) throw std::runtime_error("onnx: no graph in " + path);

    OnnxModel m;
    std::vector<OnnxValueInfo> ins;
    while (next(graph, f)) {
        if (f.wt != 2) continue;
        switch (f.no) {
            case 1: m.nodes.push_back(read_node(f.sub)); break;
            case 5: { OnnxTensor t = read_tensor(f.sub); std::string nm = t.name; m.inits[nm] = std::move(t); break; }
            case 11: ins.push_back(read_value_info(f.sub)); break;
            case 12: m.outputs.push_back(read_value_info(f.sub)); break;
            default: break;
        }
    }
    for (auto& vi : ins) if (!m.inits.count(vi.name)) m.inputs.push_back(vi);
    if (m.inputs.empty() || m.outputs.empty() || m.nodes.empty())
        throw std::runtime_error("onnx: graph has no inputs/outputs/nodes: " + path);
    return m;
}

}  // namespace fh
