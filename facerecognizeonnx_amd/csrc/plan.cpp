// plan.cpp — fusion + layout planning.  See plan.h.
//
// Fusions (each only when the intermediate value has exactly one consumer):
//   Conv/Gemm -> BatchNormalization          folded into weights/bias (fp64 on the host)
//   Conv -> Relu | PRelu | Sigmoid           epilogue activation
//   Conv (-> act) -> Add(other)              epilogue residual (order: bias, act, + residual)
//   ... Add(conv, Resize_nearest_x2(t))      residual read through a 2x nearest up-sampling
//   X -> BatchNormalization -> zero-padded Conv   (IResNet `bn1`): cannot be
//        folded into the conv exactly (border taps see 0, not the BN shift), so the BN is
//        emitted as a SECOND output of the kernel that produces X.
//   Transpose(0,2,3,1) / Reshape / Flatten   views of the channels-last storage; the Gemm
//        after an NCHW Flatten gets its K axis permuted instead.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>

namespace fh {
namespace {

struct GNode {
    std::string op;
    std::vector<std::string> in;
    std::string out;
    const OnnxNode* src = nullptr;
    bool dead = false;
    int order = 0;
    // conv / gemm payload
    std::vector<double> w;                 // ONNX layout
    std::vector<double> b;
    int Cout = 0, CinG = 0, ks = 1, stride = 1, pad = 0, group = 1;
    // fused
    Act act = Act::NONE;
    std::vector<float> slope;
    std::string res;
    ResMode res_mode = ResMode::NONE;
    std::string out2;
    std::vector<float> s2, t2;
    bool write_out1 = true;
};

[[noreturn]] void fail(const std::string& msg) { throw std::runtime_error("plan: " + msg); }

const OnnxTensor& init_of(const OnnxModel& m, const std::string& name) {
    auto it = m.inits.find(name);
    if (it == m.inits.end()) fail("expected initializer '" + name + "'");
    return it->second;
}

// Per-channel scale / shift of a BatchNormalization node — or of an element-wise Mul / Add / Sub / Div whose other operand is a
// constant (a scalar or one value per channel: the Scale layers of SCRFD's regression branches, mean / std normalisation nodes
// some exporters leave in the graph).  `channels` > 0 broadcasts a scalar constant.
void bn_scale_shift(const OnnxModel& m, const OnnxNode& bn, std::vector<double>& s, std::vector<double>& t, int channels = 0) {
    if (bn.op != "BatchNormalization") {
        const bool c_first = m.inits.count(bn.inputs.at(0)) != 0;
        const auto& c = init_of(m, bn.inputs.at(c_first ? 0 : 1)).f;
        if (c.empty()) fail(bn.op + " with an empty constant");
        if (c_first && (bn.op == "Sub" || bn.op == "Div")) fail(bn.op + " of a constant by a tensor is not supported");
        const size_t nc = c.size() == 1 && channels > 0 ? (size_t)channels : c.size();
        s.assign(nc, 1.0); t.assign(nc, 0.0);
        for (size_t i = 0; i < nc; ++i) {
            const double v = c[c.size() == 1 ? 0 : i];
            if (bn.op == "Mul") s[i] = v;
            else if (bn.op == "Div") s[i] = 1.0 / v;
            else if (bn.op == "Add") t[i] = v;
            else t[i] = -v;                                    // Sub
        }
        return;
    }
    const auto& g = init_of(m, bn.inputs[1]).f;
    const auto& be = init_of(m, bn.inputs[2]).f;
    const auto& mu = init_of(m, bn.inputs[3]).f;
    const auto& var = init_of(m, bn.inputs[4]).f;
    const double eps = bn.attr_f("epsilon", 1e-5f);
    size_t c = g.size();
    if (be.size() != c || mu.size() != c || var.size() != c) fail("BatchNormalization parameter size mismatch");
    s.resize(c); t.resize(c);
    for (size_t i = 0; i < c; ++i) {
        s[i] = (double)g[i] / std::sqrt((double)var[i] + eps);
        t[i] = (double)be[i] - (double)mu[i] * s[i];
    }
}

// how an ONNX value maps onto stored tensors
struct Val {
    int tensor = -1;
    enum Kind { NCHW4D, NHWC4D, FLAT_NCHW, FLAT_STORAGE } kind = NCHW4D;
    int rows = 0, cols = 0;     // FLAT_STORAGE view (per image)
};

}  // namespace

// Identity and (inference-mode) Dropout nodes on tensors are pass-throughs: drop them, renaming their outputs to their inputs in
// every later node and in the graph outputs.  Only the node list is copied (the initializers are shared by reference).
//
// The same renaming covers an Identity whose input is an INITIALIZER: torch's exporter de-duplicates equal parameter tensors
// (default-initialised BatchNorm / PReLU parameters, equal biases) into `Identity(initializer) -> alias` nodes, and ONNX Runtime
// (reference src/face_detector.cpp:24-26) resolves those like any other node — every later use of the alias is pointed at the
// initializer itself.  `Constant` nodes that carry a tensor (`value`) become initializers under their output name (`consts`), so that a
// weight, slope or BatchNorm vector produced by a Constant is found by init_of() as well; scalar attribute forms (value_float /
// value_int / value_ints / value_floats, opset >= 12) are turned into tensors first.
static bool strip_passthrough(const OnnxModel& m, std::vector<OnnxNode>& nodes, std::vector<OnnxValueInfo>& outputs,
                              std::map<std::string, OnnxTensor>& consts) {
    bool any = false;
    for (const auto& n : m.nodes) any = any || n.op == "Identity" || n.op == "Dropout" || n.op == "Constant";
    if (!any) return false;
    std::map<std::string, std::string> alias;
    auto A = [&](const std::string& v) { auto it = alias.find(v); return it == alias.end() ? v : it->second; };
    for (const auto& n : m.nodes) {
        if ((n.op == "Identity" || n.op == "Dropout") && !n.inputs.empty() && !n.outputs.empty()) {
            alias[n.outputs[0]] = A(n.inputs[0]);           // (a Dropout's optional mask output is never used at inference)
            continue;
        }
        if (n.op == "Constant" && !n.outputs.empty()) {
            OnnxTensor t;
            bool ok = true;
            if (n.attrs.count("value")) t = n.attrs.at("value").t;
            else if (n.attrs.count("value_float")) { t.dtype = 1; t.f = {n.attrs.at("value_float").f}; }
            else if (n.attrs.count("value_int")) { t.dtype = 7; t.i = {n.attrs.at("value_int").i}; }
            else if (n.attrs.count("value_floats")) { t.dtype = 1; t.f = n.attrs.at("value_floats").floats; t.dims = {(int64_t)t.f.size()}; }
            else if (n.attrs.count("value_ints")) { t.dtype = 7; t.i = n.attrs.at("value_ints").ints; t.dims = {(int64_t)t.i.size()}; }
            else ok = false;
            if (ok) { t.name = n.outputs[0]; consts[n.outputs[0]] = std::move(t); continue; }
        }
        OnnxNode c = n;
        for (auto& i : c.inputs) i = A(i);
        nodes.push_back(std::move(c));
    }
    outputs = m.outputs;
    for (auto& o : outputs) o.name = A(o.name);
    return true;
}

static Plan build_plan_impl(const OnnxModel& m, int inH, int inW);

Plan build_plan(const OnnxModel& m0, int inH, int inW) {
    std::vector<OnnxNode> nodes;
    std::vector<OnnxValueInfo> outputs;
    std::map<std::string, OnnxTensor> consts;
    if (!strip_passthrough(m0, nodes, outputs, consts)) return build_plan_impl(m0, inH, inW);
    OnnxModel m;                                             // same graph without the pass-through / Constant nodes
    m.nodes = std::move(nodes);
    m.outputs = std::move(outputs);
    m.inputs = m0.inputs;
    m.inits = m0.inits;                                      // (copy: only taken for graphs that contain such nodes)
    for (auto& kv : consts) m.inits[kv.first] = std::move(kv.second);
    Plan P = build_plan_impl(m, inH, inW);
    for (size_t i = 0; i < P.outputs.size() && i < m0.outputs.size(); ++i) P.outputs[i].name = m0.outputs[i].name;   // keep the file's output names
    return P;
}

static Plan build_plan_impl(const OnnxModel& m, int inH, int inW) {
    if (m.inputs.size() != 1) fail("expected exactly one graph input");
    // ---------------------------------------------------------------- 0. shape / constant pre-pass
    // Exports with dynamic axes (the public det_500m.onnx has H and W dynamic) compute Resize sizes and
    // Reshape shapes with small integer sub-graphs (Shape, Gather, Slice, Concat, Unsqueeze, Cast,
    // Mul, Div, ...).  With the input size fixed at load time they are constants: fold them here, in
    // file order (ONNX files are topologically sorted), together with the C/H/W of every 4-D value.
    struct Dim { int c = 0, h = 0, w = 0; };
    std::map<std::string, Dim> dims;
    std::map<std::string, std::vector<double>> cvals;
    std::set<std::string> shape_nodes;                    // outputs of folded nodes (never tensors)
    auto get_const = [&](const std::string& name, std::vector<double>& out) -> bool {
        auto it = cvals.find(name);
        if (it != cvals.end()) { out = it->second; return true; }
        auto ii = m.inits.find(name);
        if (ii == m.inits.end() || ii->second.numel() > 64) return false;
        out.clear();
        if (!ii->second.i.empty()) out.assign(ii->second.i.begin(), ii->second.i.end());
        else out.assign(ii->second.f.begin(), ii->second.f.end());
        return true;
    };
    dims[m.inputs[0].name] = Dim{3, inH, inW};
    for (const auto& n : m.nodes) {
        if (n.outputs.empty()) continue;
        const std::string& out = n.outputs[0];
        auto in_dim = [&](size_t k) -> const Dim* {
            if (k >= n.inputs.size()) return nullptr;
            auto it = dims.find(n.inputs[k]);
            return it == dims.end() ? nullptr : &it->second;
        };
        std::vector<double> a, b;
        if (n.op == "Conv") {
            const Dim* d = in_dim(0);
            auto wi = m.inits.find(n.inputs.size() > 1 ? n.inputs[1] : "");
            if (d && wi != m.inits.end() && wi->second.dims.size() == 4) {
                const int k = (int)wi->second.dims[2];
                auto st = n.attr_ints("strides"); auto pd = n.attr_ints("pads");
                const int s_ = st.empty() ? 1 : (int)st[0], p_ = pd.empty() ? 0 : (int)pd[0];
                dims[out] = Dim{(int)wi->second.dims[0], (d->h + 2 * p_ - k) / s_ + 1, (d->w + 2 * p_ - k) / s_ + 1};
            }
        } else if (n.op == "Resize" || n.op == "Upsample") {
            const Dim* d = in_dim(0);
            if (d) {
                Dim o = *d;
                bool ok = false;
                if (n.op == "Upsample") { if (get_const(n.inputs.at(1), a) && a.size() == 4) { o.h = (int)std::floor(d->h * a[2]); o.w = (int)std::floor(d->w * a[3]); ok = true; } }
                else {
                    if (n.inputs.size() > 3 && !n.inputs[3].empty() && get_const(n.inputs[3], a) && a.size() == 4) { o.h = (int)a[2]; o.w = (int)a[3]; ok = true; }
                    else if (n.inputs.size() > 2 && !n.inputs[2].empty() && get_const(n.inputs[2], a) && a.size() == 4) { o.h = (int)std::floor(d->h * a[2]); o.w = (int)std::floor(d->w * a[3]); ok = true; }
                    else if (n.inputs.size() == 2 && get_const(n.inputs[1], a) && a.size() == 4) { o.h = (int)std::floor(d->h * a[2]); o.w = (int)std::floor(d->w * a[3]); ok = true; }   // opset 10
                }
                if (ok) dims[out] = o;
            }
        } else if (n.op == "Shape") {
            const Dim* d = in_dim(0);
            if (d) { cvals[out] = {1.0, (double)d->c, (double)d->h, (double)d->w}; shape_nodes.insert(out); }
        } else if (n.op == "Constant") {
            auto it = n.attrs.find("value");
            if (it != n.attrs.end() && it->second.t.numel() <= 64) {
                std::vector<double> v;
                if (!it->second.t.i.empty()) v.assign(it->second.t.i.begin(), it->second.t.i.end());
                else v.assign(it->second.t.f.begin(), it->second.t.f.end());
                cvals[out] = v; shape_nodes.insert(out);
            }
        } else if (n.op == "Gather") {
            if (get_const(n.inputs.at(0), a) && get_const(n.inputs.at(1), b) && n.attr_i("axis", 0) == 0) {
                std::vector<double> v;
                for (double ix : b) { long q = (long)ix; if (q < 0) q += (long)a.size(); if (q < 0 || q >= (long)a.size()) fail("Gather index out of range"); v.push_back(a[(size_t)q]); }
                cvals[out] = v; shape_nodes.insert(out);
            }
        } else if (n.op == "Slice") {
            if (get_const(n.inputs.at(0), a)) {
                std::vector<double> st, en, ax, sp;
                bool ok = true;
                if (n.inputs.size() >= 3) {
                    ok = get_const(n.inputs[1], st) && get_const(n.inputs[2], en);
                    if (n.inputs.size() > 3 && !n.inputs[3].empty()) ok = ok && get_const(n.inputs[3], ax);
                    if (n.inputs.size() > 4 && !n.inputs[4].empty()) ok = ok && get_const(n.inputs[4], sp);
                } else {
                    for (auto v : n.attr_ints("starts")) st.push_back((double)v);
                    for (auto v : n.attr_ints("ends")) en.push_back((double)v);
                    for (auto v : n.attr_ints("axes")) ax.push_back((double)v);
                }
                if (ok && st.size() == 1 && en.size() == 1 && (ax.empty() || ax[0] == 0) && (sp.empty() || sp[0] == 1)) {
                    long L = (long)a.size(), s0 = (long)st[0], e0 = (long)std::min<double>(en[0], 1e9);
                    if (s0 < 0) s0 += L;
                    if (e0 < 0) e0 += L;
                    s0 = std::max(0L, std::min(L, s0)); e0 = std::max(0L, std::min(L, e0));
                    cvals[out] = std::vector<double>(a.begin() + s0, a.begin() + std::max(s0, e0)); shape_nodes.insert(out);
                }
            }
        } else if (n.op == "Concat") {
            std::vector<double> v;
            bool ok = !n.inputs.empty();
            for (auto& in : n.inputs) { if (!get_const(in, a) || dims.count(in)) { ok = false; break; } v.insert(v.end(), a.begin(), a.end()); }
            if (ok) { cvals[out] = v; shape_nodes.insert(out); }
        } else if (n.op == "Unsqueeze" || n.op == "Squeeze" || n.op == "Cast" || n.op == "Identity" || n.op == "Floor" || n.op == "Ceil") {
            if (!dims.count(n.inputs.at(0)) && get_const(n.inputs[0], a)) {
                if (n.op == "Floor" || (n.op == "Cast" && n.attr_i("to", 1) != 1 && n.attr_i("to", 1) != 11)) for (auto& v : a) v = std::floor(v);
                if (n.op == "Ceil") for (auto& v : a) v = std::ceil(v);
                cvals[out] = a; shape_nodes.insert(out);
            } else if (n.op == "Identity" && in_dim(0)) dims[out] = *in_dim(0);
        } else if ((n.op == "Mul" || n.op == "Div" || n.op == "Add" || n.op == "Sub") && n.inputs.size() == 2 &&
                   !dims.count(n.inputs[0]) && !dims.count(n.inputs[1]) && get_const(n.inputs[0], a) && get_const(n.inputs[1], b)) {
            const size_t L = std::max(a.size(), b.size());
            if ((a.size() == L || a.size() == 1) && (b.size() == L || b.size() == 1)) {
                std::vector<double> v(L);
                for (size_t k = 0; k < L; ++k) {
                    const double x = a[a.size() == 1 ? 0 : k], y = b[b.size() == 1 ? 0 : k];
                    v[k] = n.op == "Mul" ? x * y : n.op == "Div" ? x / y : n.op == "Add" ? x + y : x - y;
                }
                cvals[out] = v; shape_nodes.insert(out);
            }
        } else if (n.op == "Transpose" || n.op == "Reshape" || n.op == "Flatten" || n.op == "Gemm") {
            // leaves the 4-D NCHW domain: no Dim entry
        } else if (in_dim(0)) {
            dims[out] = *in_dim(0);                           // element-wise: BN, Relu, PRelu, Sigmoid, Add, ...
        }
    }
    // ---------------------------------------------------------------- 1. node list
    std::vector<GNode> g;
    g.reserve(m.nodes.size());
    int ord = 0;
    for (const auto& n : m.nodes) {
        GNode x;
        x.op = n.op; x.src = &n; x.order = ord++;
        if (n.outputs.empty()) fail("node without output");
        x.out = n.outputs[0];
        if (shape_nodes.count(x.out)) { --ord; continue; }   // folded into a constant by the pre-pass
        if (n.op == "Conv") {
            const auto& w = init_of(m, n.inputs.at(1));
            if (w.dims.size() != 4 || w.dims[2] != w.dims[3]) fail("Conv weight must be [Cout,Cin/g,k,k]");
            x.Cout = (int)w.dims[0]; x.CinG = (int)w.dims[1]; x.ks = (int)w.dims[2];
            x.group = (int)n.attr_i("group", 1);
            auto st = n.attr_ints("strides"); auto pd = n.attr_ints("pads"); auto dl = n.attr_ints("dilations");
            x.stride = st.empty() ? 1 : (int)st[0];
            if (!st.empty() && st[0] != st[1]) fail("anisotropic stride");
            x.pad = pd.empty() ? 0 : (int)pd[0];
            for (auto p : pd) if (p != x.pad) fail("asymmetric padding");
            for (auto d : dl) if (d != 1) fail("dilation != 1");
            if (n.attr_s("auto_pad", "NOTSET") != "NOTSET") fail("auto_pad not supported");
            // dense: 3x3/p1 and 1x1/p0; grouped: additionally k x k VALID (checked against the input size when the op is emitted)
            if (!((x.ks == 3 && x.pad == 1) || (x.ks == 1 && x.pad == 0) || (x.group > 1 && x.pad == 0)))
                fail("only 3x3/p1 and 1x1/p0 convolutions are supported");
            x.w.assign(w.f.begin(), w.f.end());
            if (n.inputs.size() > 2 && !n.inputs[2].empty()) { const auto& b = init_of(m, n.inputs[2]).f; x.b.assign(b.begin(), b.end()); }
            else x.b.assign((size_t)x.Cout, 0.0);
            x.in = {n.inputs[0]};
        } else if (n.op == "MatMul") {
            // [rows, K] x constant [K, N]: a bias-less Linear (torch exports nn.Linear(bias=False) this way) -> Gemm with W^T
            const auto& w = init_of(m, n.inputs.at(1));
            if (w.dims.size() != 2) fail("MatMul: the second operand must be a 2-D initializer");
            const int K = (int)w.dims[0], N = (int)w.dims[1];
            x.op = "Gemm";
            x.Cout = N; x.CinG = K;
            x.w.resize((size_t)N * K);
            for (int k = 0; k < K; ++k)
                for (int j = 0; j < N; ++j) x.w[(size_t)j * K + k] = w.f[(size_t)k * N + j];
            x.b.assign((size_t)N, 0.0);
            x.in = {n.inputs[0]};
        } else if (n.op == "Gemm") {
            const auto& w = init_of(m, n.inputs.at(1));
            if (w.dims.size() != 2) fail("Gemm weight must be 2-D");
            if (n.attr_i("transB", 0) != 1 || n.attr_i("transA", 0) != 0 || n.attr_f("alpha", 1.f) != 1.f || n.attr_f("beta", 1.f) != 1.f)
                fail("Gemm: only alpha=beta=1, transB=1 supported");
            x.Cout = (int)w.dims[0]; x.CinG = (int)w.dims[1];
            x.w.assign(w.f.begin(), w.f.end());
            if (n.inputs.size() > 2 && !n.inputs[2].empty()) { const auto& b = init_of(m, n.inputs[2]).f; x.b.assign(b.begin(), b.end()); }
            else x.b.assign((size_t)x.Cout, 0.0);
            x.in = {n.inputs[0]};
        } else if (n.op == "BatchNormalization" || n.op == "Relu" || n.op == "Sigmoid" || n.op == "Flatten" ||
                   n.op == "Transpose" || n.op == "Reshape" || n.op == "PRelu" || n.op == "Resize" || n.op == "Upsample") {
            x.in = {n.inputs[0]};
        } else if ((n.op == "Mul" || n.op == "Add" || n.op == "Sub" || n.op == "Div") && n.inputs.size() == 2 &&
                   (m.inits.count(n.inputs[0]) != 0) != (m.inits.count(n.inputs[1]) != 0)) {
            // tensor (op) constant: a per-channel affine — handled exactly like a BatchNormalization from here on
            x.op = "BatchNormalization";
            x.in = {m.inits.count(n.inputs[0]) ? n.inputs[1] : n.inputs[0]};
        } else if (n.op == "Add") {
            x.in = {n.inputs.at(0), n.inputs.at(1)};
        } else {
            fail("unsupported operator '" + n.op + "'");
        }
        g.push_back(std::move(x));
    }

    std::set<std::string> graph_outs;
    for (auto& o : m.outputs) graph_outs.insert(o.name);
    auto consumers = [&](const std::string& v) {
        int c = graph_outs.count(v) ? 1 : 0;
        for (auto& n : g) {
            if (n.dead) continue;
            for (auto& i : n.in) if (i == v) ++c;
            if (n.res == v) ++c;
        }
        return c;
    };
    auto producer = [&](const std::string& v) -> GNode* {
        for (auto& n : g) if (!n.dead && (n.out == v || (!n.out2.empty() && n.out2 == v))) return &n;
        return nullptr;
    };
    auto is_convlike = [](const GNode* n) { return n && (n->op == "Conv" || n->op == "Gemm"); };
    auto resize_is_up2 = [&](const GNode& r) {
        const OnnxNode& n = *r.src;
        if (n.attr_s("mode", "nearest") != "nearest") fail("Resize: only nearest supported");
        auto di = dims.find(n.inputs[0]);
        auto dn = dims.find(n.outputs[0]);
        if (di == dims.end() || dn == dims.end())
            fail("Resize: output size is not a load-time constant (unsupported shape sub-graph)");
        return dn->second.h == 2 * di->second.h && dn->second.w == 2 * di->second.w && dn->second.c == di->second.c;
    };

    // ---------------------------------------------------------------- 2. Conv/Gemm -> BN folding
    for (auto& bn : g) {
        if (bn.dead || bn.op != "BatchNormalization") continue;
        GNode* p = producer(bn.in[0]);
        if (!is_convlike(p) || p->out != bn.in[0] || consumers(p->out) != 1) continue;
        if (p->act != Act::NONE || !p->res.empty() || !p->out2.empty()) continue;
        std::vector<double> s, t;
        bn_scale_shift(m, *bn.src, s, t, p->Cout);
        if ((int)s.size() != p->Cout) fail("BN channel mismatch after " + p->op);
        size_t per = p->w.size() / (size_t)p->Cout;
        for (int co = 0; co < p->Cout; ++co) {
            for (size_t k = 0; k < per; ++k) p->w[(size_t)co * per + k] *= s[co];
            p->b[co] = p->b[co] * s[co] + t[co];
        }
        p->out = bn.out;
        bn.dead = true;
    }
    // ---------------------------------------------------------------- 3. activation fusion
    for (auto& a : g) {
        if (a.dead || !(a.op == "Relu" || a.op == "PRelu" || a.op == "Sigmoid")) continue;
        GNode* p = producer(a.in[0]);
        if (!p || p->op != "Conv" || p->out != a.in[0] || consumers(p->out) != 1) continue;
        if (p->act != Act::NONE || !p->res.empty() || !p->out2.empty()) continue;
        if (a.op == "Relu") p->act = Act::RELU;
        else if (a.op == "Sigmoid") p->act = Act::SIGMOID;
        else {
            const auto& sl = init_of(m, a.src->inputs.at(1)).f;
            if (sl.size() == 1) p->slope.assign((size_t)p->Cout, sl[0]);
            else if ((int)sl.size() == p->Cout) p->slope = sl;
            else fail("PRelu slope size mismatch");
            p->act = Act::PRELU;
        }
        p->out = a.out;
        a.dead = true;
    }
    // ---------------------------------------------------------------- 4. residual (Add) fusion
    for (auto& add : g) {
        if (add.dead || add.op != "Add") continue;
        for (int side = 0; side < 2; ++side) {
            GNode* p = producer(add.in[side]);
            if (!p || p->op != "Conv" || p->group != 1 || p->out != add.in[side]) continue;
            if (consumers(p->out) != 1 || !p->res.empty() || !p->out2.empty()) continue;
            const std::string other = add.in[1 - side];
            if (other == p->out) continue;
            GNode* q = producer(other);
            if (q && (q->op == "Resize" || q->op == "Upsample") && consumers(q->out) == 1 && resize_is_up2(*q)) {
                p->res = q->in[0]; p->res_mode = ResMode::UP2X; q->dead = true;
            } else {
                p->res = other; p->res_mode = ResMode::SAME;
            }
            p->out = add.out;
            add.dead = true;
            break;
        }
    }
    // ---------------------------------------------------------------- 5. BN as second output
    for (auto& bn : g) {
        if (bn.dead || bn.op != "BatchNormalization") continue;
        GNode* p = producer(bn.in[0]);
        if (!p || p->op != "Conv" || p->group != 1 || p->out != bn.in[0] || !p->out2.empty()) continue;
        std::vector<double> s, t;
        bn_scale_shift(m, *bn.src, s, t, p->Cout);
        if ((int)s.size() != p->Cout) fail("BN channel mismatch (second output)");
        p->s2.assign(s.begin(), s.end());
        p->t2.assign(t.begin(), t.end());
        p->out2 = bn.out;
        bn.dead = true;
        if (consumers(p->out) == 0) p->write_out1 = false;
    }

    // ---------------------------------------------------------------- 6. topological order
    std::vector<GNode*> live;
    for (auto& n : g) if (!n.dead) live.push_back(&n);
    std::vector<GNode*> sorted;
    {
        std::set<std::string> ready;
        ready.insert(m.inputs[0].name);
        std::vector<bool> done(live.size(), false);
        for (size_t iter = 0; iter < live.size(); ++iter) {
            bool progressed = false;
            for (size_t i = 0; i < live.size(); ++i) {
                if (done[i]) continue;
                GNode* n = live[i];
                bool ok = true;
                for (auto& v : n->in) if (!ready.count(v)) ok = false;
                if (!n->res.empty() && !ready.count(n->res)) ok = false;
                if (!ok) continue;
                done[i] = true; progressed = true;
                sorted.push_back(n);
                ready.insert(n->out);
                if (!n->out2.empty()) ready.insert(n->out2);
                break;               // restart scan → stable w.r.t. file order
            }
            if (!progressed) fail("graph is not a DAG over the supported ops (unresolved input)");
        }
    }

    // ---------------------------------------------------------------- 7. shapes, views, ops
    Plan P;
    P.inH = inH; P.inW = inW;
    std::map<std::string, Val> vals;
    auto new_tensor = [&](const std::string& name, int H, int W, int C) {
        PTensor t; t.name = name; t.H = H; t.W = W; t.C = C;
        P.tensors.push_back(t);
        return (int)P.tensors.size() - 1;
    };
    {
        const auto& shp = m.inputs[0].shape;
        if (shp.size() != 4) fail("graph input must be 4-D NCHW");
        if (shp[1] > 0 && shp[1] != 3) fail("graph input must have 3 channels");
        P.input = new_tensor(m.inputs[0].name, inH, inW, 4);      // RGB + one zero lane
        P.tensors[P.input].is_input = true;
        Val v; v.tensor = P.input; v.kind = Val::NCHW4D;
        vals[m.inputs[0].name] = v;
    }
    auto need4d = [&](const std::string& name) -> const Val& {
        auto it = vals.find(name);
        if (it == vals.end()) fail("value '" + name + "' used before definition");
        if (it->second.kind != Val::NCHW4D) fail("operator needs a 4-D NCHW value: '" + name + "'");
        return it->second;
    };

    for (GNode* n : sorted) {
        if (n->op == "Conv") {
            const Val& vi = need4d(n->in[0]);
            const PTensor ti = P.tensors[vi.tensor];
            const int logicalCin = n->CinG * n->group;
            const bool is_input = vi.tensor == P.input;
            if (!(ti.C == logicalCin || (is_input && logicalCin == 3))) fail("Conv input channel mismatch at " + n->out);
            POp op;
            op.name = n->src->name.empty() ? n->out : n->src->name;
            op.in = vi.tensor; op.ks = n->ks; op.stride = n->stride; op.pad = n->pad;
            op.H = ti.H; op.W = ti.W; op.Cin = ti.C; op.Cout = n->Cout;
            op.Ho = (ti.H + 2 * n->pad - n->ks) / n->stride + 1;
            op.Wo = (ti.W + 2 * n->pad - n->ks) / n->stride + 1;
            op.act = n->act; op.slope = n->slope;
            op.bias.assign(n->b.begin(), n->b.end());
            const int taps = n->ks * n->ks;
            if (n->group == 1) {
                op.kind = OpKind::CONV;
                op.weight.assign((size_t)n->Cout * taps * ti.C, 0.f);
                for (int co = 0; co < n->Cout; ++co)
                    for (int ci = 0; ci < logicalCin; ++ci)
                        for (int t = 0; t < taps; ++t)
                            op.weight[((size_t)co * taps + t) * ti.C + ci] =
                                (float)n->w[((size_t)co * logicalCin + ci) * taps + t];
                op.macs = (double)op.Ho * op.Wo * n->Cout * taps * logicalCin;
            } else {
                const bool depthwise = n->group == logicalCin && n->CinG == 1 && n->Cout == logicalCin;
                const int coutG = n->Cout / n->group;
                if (ti.C % 4 || n->Cout % 4) fail("grouped Conv needs C % 4 == 0");
                if (!n->res.empty() || !n->out2.empty()) fail("internal: fused residual on a grouped conv");
                if (depthwise && n->ks == 3 && n->pad == 1) {
                    op.kind = OpKind::DWCONV;
                    op.weight.assign((size_t)9 * ti.C, 0.f);
                    for (int c = 0; c < ti.C; ++c)
                        for (int t = 0; t < 9; ++t) op.weight[(size_t)t * ti.C + c] = (float)n->w[(size_t)c * 9 + t];
                    op.macs = (double)op.Ho * op.Wo * ti.C * 9;
                } else if (depthwise && n->pad == 0 && n->stride == 1 && n->ks == ti.H && n->ks == ti.W) {
                    op.kind = OpKind::DWGLOBAL;                  // k x k VALID over a k x k map -> 1 x 1
                    op.weight.assign((size_t)taps * ti.C, 0.f);
                    for (int c = 0; c < ti.C; ++c)
                        for (int t = 0; t < taps; ++t) op.weight[(size_t)t * ti.C + c] = (float)n->w[(size_t)c * taps + t];
                    op.macs = (double)ti.C * taps;
                } else if (n->ks == 3 && n->pad == 1 && n->CinG == coutG && (coutG == 2 || coutG == 4) && n->Cout == logicalCin) {
                    op.kind = OpKind::GCONV;                     // weight [9][Cout][G]: tap-major, then output channel, then its G inputs
                    const int G = coutG;
                    op.weight_group = G;
                    op.weight.assign((size_t)9 * n->Cout * G, 0.f);
                    for (int co = 0; co < n->Cout; ++co)
                        for (int j = 0; j < G; ++j)
                            for (int t = 0; t < 9; ++t)
                                op.weight[((size_t)t * n->Cout + co) * G + j] = (float)n->w[((size_t)co * G + j) * 9 + t];
                    op.macs = (double)op.Ho * op.Wo * n->Cout * 9 * G;
                } else if ((n->ks == 3 && n->pad == 1) || (n->ks == 1 && n->pad == 0)) {
                    // any other grouping: a dense convolution with block-diagonal weights (correct, not fast)
                    op.kind = OpKind::CONV;
                    op.weight.assign((size_t)n->Cout * taps * ti.C, 0.f);
                    for (int co = 0; co < n->Cout; ++co)
                        for (int j = 0; j < n->CinG; ++j)
                            for (int t = 0; t < taps; ++t)
                                op.weight[((size_t)co * taps + t) * ti.C + (co / coutG) * n->CinG + j] = (float)n->w[((size_t)co * n->CinG + j) * taps + t];
                    op.macs = (double)op.Ho * op.Wo * n->Cout * taps * n->CinG;
                } else {
                    fail("unsupported grouped Conv geometry at " + n->out);
                }
            }
            if (op.kind == OpKind::CONV && ti.C % 4) fail("Conv needs stored Cin % 4 == 0 at " + n->out);
            if (n->write_out1) {
                op.out = new_tensor(n->out, op.Ho, op.Wo, n->Cout);
                Val v; v.tensor = op.out; vals[n->out] = v;
            }
            if (!n->out2.empty()) {
                op.out2 = new_tensor(n->out2, op.Ho, op.Wo, n->Cout);
                op.s2 = n->s2; op.t2 = n->t2;
                Val v; v.tensor = op.out2; vals[n->out2] = v;
            }
            if (!n->res.empty()) {
                const Val& vr = need4d(n->res);
                const PTensor& tr = P.tensors[vr.tensor];
                op.res = vr.tensor; op.res_mode = n->res_mode;
                const int f = n->res_mode == ResMode::UP2X ? 2 : 1;
                if (tr.C != n->Cout || tr.H * f != op.Ho || tr.W * f != op.Wo) fail("residual shape mismatch at " + n->out);
            }
            P.ops.push_back(std::move(op));
        } else if (n->op == "Gemm") {
            auto it = vals.find(n->in[0]);
            if (it == vals.end()) fail("Gemm input undefined");
            const Val vi = it->second;
            const PTensor ti = P.tensors[vi.tensor];
            const int K = n->CinG;
            if ((size_t)K != ti.elems()) fail("Gemm K does not match the flattened producer");
            POp op; op.kind = OpKind::GEMM; op.name = n->src->name.empty() ? n->out : n->src->name;
            op.in = vi.tensor; op.ks = 1; op.stride = 1; op.pad = 0; op.H = op.W = op.Ho = op.Wo = 1;
            op.Cin = K; op.Cout = n->Cout;
            op.bias.assign(n->b.begin(), n->b.end());
            op.weight.resize((size_t)n->Cout * K);
            const int HW = ti.H * ti.W, Cc = ti.C;
            const bool permute = vi.kind == Val::FLAT_NCHW && HW > 1;
            if (!(vi.kind == Val::FLAT_NCHW || vi.kind == Val::FLAT_STORAGE)) fail("Gemm expects a flattened input");
            for (int o = 0; o < n->Cout; ++o)
                for (int k = 0; k < K; ++k) {
                    int src = k;                               // storage index k = hw*C + c
                    if (permute) { int hw = k / Cc, c = k % Cc; src = c * HW + hw; }
                    op.weight[(size_t)o * K + k] = (float)n->w[(size_t)o * K + src];
                }
            op.macs = (double)K * n->Cout;
            op.out = new_tensor(n->out, 1, 1, n->Cout);
            Val v; v.tensor = op.out; v.kind = Val::FLAT_STORAGE; v.rows = 1; v.cols = n->Cout;
            vals[n->out] = v;
            P.ops.push_back(std::move(op));
        } else if (n->op == "BatchNormalization") {
            auto it = vals.find(n->in[0]);
            if (it == vals.end()) fail("BN input undefined");
            const Val vi = it->second;
            const PTensor ti = P.tensors[vi.tensor];
            std::vector<double> s, t;
            bn_scale_shift(m, *n->src, s, t, vi.kind == Val::FLAT_STORAGE ? vi.cols : (vi.tensor == P.input ? 3 : ti.C));
            POp op; op.kind = OpKind::AFFINE; op.name = n->out; op.in = vi.tensor;
            op.H = op.Ho = ti.H; op.W = op.Wo = ti.W; op.Cin = op.Cout = ti.C;
            if (vi.kind == Val::NCHW4D) {
                if (!((int)s.size() == ti.C || (vi.tensor == P.input && s.size() == 3))) fail("standalone BN channel mismatch");
                s.resize(ti.C, 0.0); t.resize(ti.C, 0.0);
            } else {
                if ((int)s.size() != (vi.kind == Val::FLAT_STORAGE ? vi.cols : ti.C) || ti.H * ti.W != 1) fail("standalone BN on a flattened value needs H=W=1");
            }
            op.s2.assign(s.begin(), s.end()); op.t2.assign(t.begin(), t.end());
            op.out = new_tensor(n->out, ti.H, ti.W, ti.C);
            Val v = vi; v.tensor = op.out; vals[n->out] = v;
            P.ops.push_back(std::move(op));
        } else if (n->op == "Relu" || n->op == "Sigmoid" || n->op == "PRelu") {
            const Val& vi = need4d(n->in[0]);
            const PTensor ti = P.tensors[vi.tensor];
            POp op; op.kind = OpKind::ACT; op.name = n->out; op.in = vi.tensor;
            op.H = op.Ho = ti.H; op.W = op.Wo = ti.W; op.Cin = op.Cout = ti.C;
            op.act = n->op == "Relu" ? Act::RELU : n->op == "Sigmoid" ? Act::SIGMOID : Act::PRELU;
            if (op.act == Act::PRELU) {
                const auto& sl = init_of(m, n->src->inputs.at(1)).f;
                if (sl.size() == 1) op.slope.assign((size_t)ti.C, sl[0]); else op.slope = sl;
                op.slope.resize(ti.C, 0.f);
            }
            op.out = new_tensor(n->out, ti.H, ti.W, ti.C);
            Val v; v.tensor = op.out; vals[n->out] = v;
            P.ops.push_back(std::move(op));
        } else if (n->op == "Add") {
            const Val va = need4d(n->in[0]);
            const Val vb = need4d(n->in[1]);
            const PTensor ta = P.tensors[va.tensor], tb = P.tensors[vb.tensor];
            if (ta.H != tb.H || ta.W != tb.W || ta.C != tb.C) fail("Add operands differ in shape (broadcast unsupported)");
            POp op; op.kind = OpKind::ADD; op.name = n->out; op.in = va.tensor; op.in2 = vb.tensor;
            op.H = op.Ho = ta.H; op.W = op.Wo = ta.W; op.Cin = op.Cout = ta.C;
            op.out = new_tensor(n->out, ta.H, ta.W, ta.C);
            Val v; v.tensor = op.out; vals[n->out] = v;
            P.ops.push_back(std::move(op));
        } else if (n->op == "Resize" || n->op == "Upsample") {
            if (!resize_is_up2(*n)) fail("Resize: only x2 nearest supported");
            const Val& vi = need4d(n->in[0]);
            const PTensor ti = P.tensors[vi.tensor];
            POp op; op.kind = OpKind::UPSAMPLE; op.name = n->out; op.in = vi.tensor;
            op.H = ti.H; op.W = ti.W; op.Ho = 2 * ti.H; op.Wo = 2 * ti.W; op.Cin = op.Cout = ti.C;
            op.out = new_tensor(n->out, op.Ho, op.Wo, ti.C);
            Val v; v.tensor = op.out; vals[n->out] = v;
            P.ops.push_back(std::move(op));
        } else if (n->op == "Transpose") {
            const Val& vi = need4d(n->in[0]);
            auto perm = n->src->attr_ints("perm");
            if (perm != std::vector<int64_t>{0, 2, 3, 1}) fail("Transpose: only perm (0,2,3,1) supported");
            Val v = vi; v.kind = Val::NHWC4D;
            vals[n->out] = v;
        } else if (n->op == "Flatten") {
            auto it = vals.find(n->in[0]);
            if (it == vals.end()) fail("Flatten input undefined");
            Val v = it->second;
            if (n->src->attr_i("axis", 1) != 1) fail("Flatten: only axis=1 supported");
            if (v.kind == Val::NCHW4D) v.kind = Val::FLAT_NCHW;
            else if (v.kind == Val::NHWC4D) { v.kind = Val::FLAT_STORAGE; v.rows = 1; v.cols = (int)P.tensors[v.tensor].elems(); }
            vals[n->out] = v;
        } else if (n->op == "Reshape") {
            auto it = vals.find(n->in[0]);
            if (it == vals.end()) fail("Reshape input undefined");
            Val v = it->second;
            std::vector<double> shp;
            if (!get_const(n->src->inputs.at(1), shp)) fail("Reshape: target shape is not a load-time constant");
            if (shp.empty()) fail("Reshape: empty target shape");
            const PTensor& t = P.tensors[v.tensor];
            const int64_t last = (int64_t)shp.back();
            if (last <= 0 || t.elems() % (size_t)last) fail("Reshape: last dimension must be a positive divisor of the tensor size");
            const bool storage_order = v.kind == Val::NHWC4D || v.kind == Val::FLAT_STORAGE ||
                                       (v.kind == Val::NCHW4D && (t.H * t.W == 1 || t.C == 1)) ||
                                       (v.kind == Val::FLAT_NCHW && (t.H * t.W == 1 || t.C == 1));
            if (!storage_order) fail("Reshape of an NCHW-ordered value would need a physical transpose (not supported)");
            v.kind = Val::FLAT_STORAGE; v.cols = (int)last; v.rows = (int)(t.elems() / (size_t)last);
            vals[n->out] = v;
        } else {
            fail("internal: unhandled op " + n->op);
        }
    }

    // ---------------------------------------------------------------- 7b. horizontal merge of small sibling convs
    {
        std::vector<bool> gone(P.ops.size(), false);
        for (size_t i = 0; i < P.ops.size(); ++i) {
            POp& a = P.ops[i];
            auto eligible = [](const POp& o) {
                return o.kind == OpKind::CONV && o.res < 0 && o.out2 < 0 && o.out >= 0 && o.slope.empty() && o.outs.empty();
            };
            if (gone[i] || !eligible(a)) continue;
            std::vector<size_t> grp{i};
            int ctot = a.Cout;
            for (size_t j = i + 1; j < P.ops.size(); ++j) {
                const POp& b = P.ops[j];
                if (gone[j] || !eligible(b) || b.in != a.in || b.ks != a.ks || b.stride != a.stride || b.pad != a.pad) continue;
                if (ctot + b.Cout > 32 || grp.size() >= 3) continue;
                // the sibling must not depend on anything produced between i and j (it only reads `in`)
                grp.push_back(j);
                ctot += b.Cout;
            }
            if (grp.size() < 2) continue;
            const int taps = a.ks * a.ks, kin = taps * a.Cin;
            POp m = a;
            m.name = a.name + "+merged";
            m.weight.assign((size_t)ctot * kin, 0.f);
            m.bias.assign((size_t)ctot, 0.f);
            m.out = -1; m.act = Act::NONE; m.Cout = ctot; m.macs = 0;
            int c0 = 0;
            for (size_t g : grp) {
                const POp& b = P.ops[g];
                std::copy(b.weight.begin(), b.weight.end(), m.weight.begin() + (size_t)c0 * kin);
                std::copy(b.bias.begin(), b.bias.end(), m.bias.begin() + c0);
                m.outs.push_back(b.out); m.out_c0.push_back(c0); m.out_act.push_back((int)b.act);
                m.macs += b.macs;
                c0 += b.Cout;
                if (g != i) gone[g] = true;
            }
            m.out_c0.push_back(c0);
            P.ops[i] = std::move(m);
        }
        std::vector<POp> kept;
        for (size_t i = 0; i < P.ops.size(); ++i) if (!gone[i]) kept.push_back(std::move(P.ops[i]));
        P.ops.swap(kept);
    }

    // ---------------------------------------------------------------- 7c. depthwise 3x3 (stride 1 | 2) -> pointwise 1x1 fusion
    // (large maps only: the fused kernel works on 8x16 spatial tiles, small maps would be mostly padding)
    {
        auto uses = [&](int t) {
            int c = 0;
            for (auto& o : P.ops) for (int x : {o.in, o.in2, o.res}) if (x == t) ++c;
            for (auto& o : m.outputs) { auto it = vals.find(o.name); if (it != vals.end() && it->second.tensor == t) ++c; }
            return c;
        };
        std::vector<bool> gone(P.ops.size(), false);
        for (size_t i = 0; i < P.ops.size(); ++i) {
            POp& d = P.ops[i];
            if (d.kind != OpKind::DWCONV || (d.stride != 1 && d.stride != 2) || d.Ho * d.Wo < (d.stride == 1 ? 1600 : 6400) || uses(d.out) != 1) continue;
            for (size_t j = i + 1; j < P.ops.size(); ++j) {
                POp& c = P.ops[j];
                if (c.in != d.out) continue;
                const bool ok = c.kind == OpKind::CONV && c.ks == 1 && c.stride == 1 && c.res < 0 && c.out2 < 0 && c.outs.empty() &&
                                c.Cin % 4 == 0 && c.Cout <= 128 &&      // one n-tile: the depthwise part is never recomputed
                                (c.act == Act::NONE || c.act == Act::RELU) && (d.act == Act::NONE || d.act == Act::RELU);
                if (!ok) break;
                POp f = c;
                f.kind = OpKind::DWPW;
                f.name = d.name + "+" + c.name;
                f.in = d.in;
                f.dw_weight = d.weight; f.dw_bias = d.bias; f.dw_act = d.act; f.dw_stride = d.stride;
                f.H = d.H; f.W = d.W;
                f.macs = d.macs + c.macs;
                P.ops[j] = std::move(f);
                gone[i] = true;
                break;
            }
        }
        std::vector<POp> kept;
        for (size_t i = 0; i < P.ops.size(); ++i) if (!gone[i]) kept.push_back(std::move(P.ops[i]));
        P.ops.swap(kept);
    }

    // ---------------------------------------------------------------- 8. outputs
    for (const auto& o : m.outputs) {
        auto it = vals.find(o.name);
        if (it == vals.end()) fail("graph output '" + o.name + "' was never produced");
        const Val& v = it->second;
        const PTensor& t = P.tensors[v.tensor];
        OutDesc d; d.name = o.name; d.tensor = v.tensor;
        if (v.kind == Val::FLAT_STORAGE) { d.rows = v.rows; d.cols = v.cols; }
        else if (v.kind == Val::NHWC4D) { d.rows = t.H * t.W; d.cols = t.C; }
        else if (t.H * t.W == 1) { d.rows = 1; d.cols = t.C; }
        else fail("graph output '" + o.name + "' is NCHW-ordered; only channels-last / flattened outputs are supported");
        P.tensors[v.tensor].is_output = true;
        P.outputs.push_back(d);
    }

    // ---------------------------------------------------------------- 9. liveness + arena
    for (size_t i = 0; i < P.ops.size(); ++i) {
        const POp& op = P.ops[i];
        std::vector<int> touched{op.in, op.in2, op.res, op.out, op.out2};
        touched.insert(touched.end(), op.outs.begin(), op.outs.end());
        for (int t : touched) {
            if (t < 0) continue;
            PTensor& pt = P.tensors[t];
            if (pt.first < 0) pt.first = (int)i;
            pt.last = (int)i;
        }
    }
    // BatchNorm-in-the-transform links (POp::bn_src): the consumer may read the producer's PLAIN output at its own position,
    // which the op list above does not show — extend that tensor's lifetime so the arena cannot hand its buffer out in between
    // (e.g. when a block's shortcut convolution, the plain output's last listed reader, is scheduled before conv1).
    for (size_t i = 0; i < P.ops.size(); ++i) {
        POp& c = P.ops[i];
        if (c.kind != OpKind::CONV || c.ks != 3 || c.stride != 1 || c.pad != 1 || c.in < 0) continue;
        int uses = 0, prod = -1;
        for (size_t j = 0; j < P.ops.size(); ++j) {
            const POp& o = P.ops[j];
            for (int x : {o.in, o.in2, o.res}) if (x == c.in) ++uses;
            if (o.out2 == c.in && o.out >= 0 && o.kind == OpKind::CONV && j < i) prod = (int)j;
        }
        for (const auto& o : P.outputs) if (o.tensor == c.in) ++uses;
        if (prod < 0 || uses != 1) continue;
        c.bn_src = prod;
        PTensor& plain = P.tensors[P.ops[prod].out];
        plain.last = std::max(plain.last, (int)i);
    }
    // Shortcut-in-the-K-loop links (POp::sc_src)
    for (size_t i = 0; i < P.ops.size(); ++i) {
        POp& c = P.ops[i];
        if (c.kind != OpKind::CONV || c.ks != 3 || c.res < 0 || c.res_mode != ResMode::SAME || c.act != Act::NONE || !c.outs.empty() ||
            c.Cin % 32 != 0) continue;
        int uses = 0, prod = -1;
        for (size_t j = 0; j < P.ops.size(); ++j) {
            const POp& o = P.ops[j];
            for (int x : {o.in, o.in2, o.res}) if (x == c.res) ++uses;
            if (o.out == c.res && j < i) prod = (int)j;
        }
        for (const auto& o : P.outputs) if (o.tensor == c.res) ++uses;
        if (prod < 0 || uses != 1) continue;
        const POp& x = P.ops[prod];
        if (x.kind != OpKind::CONV || x.ks != 1 || x.pad != 0 || x.act != Act::NONE || x.out2 >= 0 || x.res >= 0 || !x.outs.empty() ||
            x.Cin % 32 != 0 || x.Cout != c.Cout || x.Ho != c.Ho || x.Wo != c.Wo || x.in < 0) continue;
        c.sc_src = prod;
        PTensor& src = P.tensors[x.in];
        src.last = std::max(src.last, (int)i);
    }
    const int nops = (int)P.ops.size();
    for (auto& t : P.tensors) {
        if (t.is_input) t.first = 0;
        if (t.is_input || t.is_output) t.last = nops;       // keep for the caller
        if (t.first < 0) { t.first = -1; t.last = -1; }         // fused away: never materialised
    }
    {
        // greedy first-fit on (lifetime, size); offsets aligned to 64 floats (256 B)
        std::vector<int> order(P.tensors.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return P.tensors[a].elems() > P.tensors[b].elems(); });
        std::vector<int> placed;
        for (int id : order) {
            PTensor& t = P.tensors[id];
            if (t.first < 0) continue;
            const size_t sz = (t.elems() + 63) / 64 * 64;
            std::vector<std::pair<size_t, size_t>> busy;
            for (int q : placed) {
                const PTensor& o = P.tensors[q];
                if (o.last < t.first || t.last < o.first) continue;
                busy.push_back({o.offset, o.offset + (o.elems() + 63) / 64 * 64});
            }
            std::sort(busy.begin(), busy.end());
            size_t off = 0;
            for (auto& b : busy) { if (off + sz <= b.first) break; off = std::max(off, b.second); }
            t.offset = off;
            P.arena_elems = std::max(P.arena_elems, off + sz);
            placed.push_back(id);
        }
    }
    // ---------------------------------------------------------------- 10. accounting
    for (auto& op : P.ops) {
        double b = 0;
        for (int t : {op.in, op.in2, op.res, op.out, op.out2}) if (t >= 0) b += (double)P.tensors[t].elems() * 4;
        for (int t : op.outs) b += (double)P.tensors[t].elems() * 4;
        op.bytes = b;
        P.macs += op.macs;
        P.act_bytes += b;
        P.weight_bytes += (double)(op.weight.size() + op.bias.size() + op.slope.size() + op.s2.size() + op.t2.size() + op.dw_weight.size() + op.dw_bias.size()) * 4;
    }
    return P;
}

std::string Plan::describe() const {
    static const char* kinds[] = {"CONV", "DWCONV", "GEMM", "AFFINE", "ACT", "ADD", "UPSAMPLE", "DW+PW", "DWGLOBAL", "GCONV"};
    static const char* acts[] = {"", "+relu", "+prelu", "+sigmoid"};
    std::ostringstream os;
    os << "input " << inH << "x" << inW << "  ops " << ops.size() << "  tensors " << tensors.size()
       << "  GMAC/image " << macs * 1e-9 << "  act MB/image " << act_bytes * 1e-6
       << "  weights MB " << weight_bytes * 1e-6 << "  arena MB/image " << arena_elems * 4e-6 << "\n";
    for (size_t i = 0; i < ops.size(); ++i) {
        const POp& o = ops[i];
        os << i << " " << kinds[(int)o.kind] << " k" << o.ks << "s" << o.stride << " " << o.H << "x" << o.W << "x" << o.Cin
           << " -> " << o.Ho << "x" << o.Wo << "x" << o.Cout << acts[(int)o.act];
        if (o.kind == OpKind::DWPW && o.dw_stride != 1) os << " (depthwise s" << o.dw_stride << ")";
        if (!o.outs.empty()) os << " [merged x" << o.outs.size() << "]";
        if (o.res >= 0) os << (o.res_mode == ResMode::UP2X ? " +res(up2x)" : " +res");
        if (o.out2 >= 0) os << (o.out >= 0 ? " +bn2nd" : " bn2nd-only");
        if (o.bn_src >= 0) os << " bn<-op" << o.bn_src;
        if (o.sc_src >= 0) os << " sc<-op" << o.sc_src;
        os << "  [in t" << o.in << " out t" << o.out << " out2 t" << o.out2 << "]";
        os << "  MMAC " << o.macs * 1e-6 << "\n";
    }
    for (auto& d : outputs) os << "out " << d.name << " [" << d.rows << "x" << d.cols << "]\n";
    for (size_t i = 0; i < tensors.size(); ++i) {                  // arena placement (per image, in floats) and lifetime in op indices
        const PTensor& t = tensors[i];
        if (t.first < 0) continue;
        os << "tensor t" << i << " " << t.H << "x" << t.W << "x" << t.C << " off " << t.offset << " live " << t.first << ".." << t.last << "\n";
    }
    return os.str();
}

}  // namespace fh
