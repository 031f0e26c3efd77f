// engine.cpp — weight upload, arena management and the launch sequences of the face path.
#include "engine.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>

namespace fh {

void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " in " + what);
}

// Layout epoch: a captured HIP graph (api.cpp, GraphCall) holds the device addresses of every buffer its kernels touched — the
// arena, the Winograd workspaces, candidate / key / crop buffers — not only the staging buffers of the call that captured it.
// Any (re)allocation or release of a DevBuf in the process makes such a graph suspect, so the epoch is part of every graph key.
static std::atomic<unsigned long long> g_layout_epoch{1};
unsigned long long layout_epoch() { return g_layout_epoch.load(std::memory_order_relaxed); }

DevBuf::~DevBuf() {
    if (p) { (void)hipFree(p); g_layout_epoch.fetch_add(1, std::memory_order_relaxed); }
}
void DevBuf::ensure(size_t n) {
    if (n <= bytes) return;
    g_layout_epoch.fetch_add(1, std::memory_order_relaxed);
    if (p) { FH_HIP(hipDeviceSynchronize()); FH_HIP(hipFree(p)); p = nullptr; bytes = 0; }
    FH_HIP(hipMalloc(&p, n));
    bytes = n;
}

// ------------------------------------------------------------------------------------------ KernelTimer
KernelTimer& KernelTimer::get() { static KernelTimer t; return t; }
hipEvent_t KernelTimer::take() {
    if (!pool_.empty()) { hipEvent_t e = pool_.back(); pool_.pop_back(); return e; }
    hipEvent_t e; FH_HIP(hipEventCreate(&e)); return e;
}
void KernelTimer::begin(hipStream_t s) {
    if (!enabled) return;
    cur_ = take();
    FH_HIP(hipEventRecord(cur_, s));
}
void KernelTimer::end(hipStream_t s, int tag, double flops, double bytes) {
    if (!enabled || !cur_) return;
    hipEvent_t b = take();
    FH_HIP(hipEventRecord(b, s));
    recs_.push_back(Rec{cur_, b, tag, flops, bytes});
    cur_ = nullptr;
}
void KernelTimer::collect(double* ms, double* flops, double* bytes, long long* launches) {
    for (int i = 0; i < kTags; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; launches[i] = 0; }
    for (auto& r : recs_) {
        FH_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        FH_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.tag] += t; flops[r.tag] += r.flops; bytes[r.tag] += r.bytes; launches[r.tag] += 1;
        pool_.push_back(r.a); pool_.push_back(r.b);
    }
    recs_.clear();
}

int KernelTimer::collect_ops(double* ms, double* flops, int* tag, int cap) {
    int n = 0;
    for (auto& r : recs_) {
        FH_HIP(hipEventSynchronize(r.b));
        float t = 0.f;
        FH_HIP(hipEventElapsedTime(&t, r.a, r.b));
        if (n < cap) { ms[n] = t; flops[n] = r.flops; tag[n] = r.tag; ++n; }
        pool_.push_back(r.a); pool_.push_back(r.b);
    }
    recs_.clear();
    return n;
}

static bool debug_sync() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_DEBUG_SYNC"); v = e ? atoi(e) : 0; }
    return v != 0;
}

// ------------------------------------------------------------------------------------------ Net

// Winograd pays from 128 input channels on (IResNet-50, B = 128: 14.13 ms direct, 11.2 ms with >= 256, 10.5 ms with >= 128,
// 10.6 ms with >= 64: below 128 the two transform passes cost what the matrix cores save)
static constexpr int kWinoMinCin = 128;
static constexpr long kWinoMinTiles = 256;      // 4x4-output tiles per launch below which the direct form is used
static long wino2_min_blocks() {                // workgroups per launch below which conv_wino2.hip's fused F(2x2) form is not used (FACEHIP_WINO2=0: never)
    static long v = -1;
    if (v < 0) {
        const char* e = getenv("FACEHIP_WINO2");
        const char* m = getenv("FACEHIP_WINO2_MIN");
        v = e && atoi(e) == 0 ? (1L << 60) : m ? atol(m) : 64;      // (measured, IResNet-50: fused form ahead from B = 8 on, level at B = 4)
    }
    return v;
}

Net::Net(const std::string& onnx_path, int default_h, int default_w) {
    OnnxModel m = load_onnx(onnx_path);
    // reference src/face_detector.cpp:39-57: adopt the model's static H/W when > 0, else keep defaults
    int H = default_h, W = default_w;
    const auto& shp = m.inputs[0].shape;
    if (shp.size() == 4) {
        if (shp[2] > 0) H = (int)shp[2];
        if (shp[3] > 0) W = (int)shp[3];
    }
    plan_ = build_plan(m, H, W);

    std::vector<float> host;
    auto push = [&](const float* src, size_t n) {
        size_t off = (host.size() + 63) / 64 * 64;
        host.resize(off + n, 0.f);
        if (n) memcpy(host.data() + off, src, n * sizeof(float));
        return off;
    };
    dev_.resize(plan_.ops.size());
    // IResNet's pre-conv BatchNorm in front of a DIRECT 3x3 convolution (Cin < 128: never Winograd): conv(W, s*x + t) with zero padding
    // = conv(W*s, x) + sum over the taps that are inside the image of W[tap]*t — the shift term only depends on which border the
    // output pixel touches, so it becomes one bias vector per border class (9 of them) and the producer's normalised second output
    // (a full extra activation write: 411 MB behind the stem at B = 128) is never needed.  Exact up to fp32 rounding.
    for (size_t i = 0; i < plan_.ops.size(); ++i) {
        POp& c = plan_.ops[i];
        if (c.bn_src < 0 || c.kind != OpKind::CONV || c.ks != 3 || c.stride != 1 || c.pad != 1 || c.H < 2 || c.W < 2) continue;
        const bool wino_ok = c.Cin >= kWinoMinCin && c.Cin % 32 == 0 && c.Cout % 4 == 0 && c.outs.empty() && c.res_mode != ResMode::UP2X;
        const POp& pr = plan_.ops[c.bn_src];
        if (wino_ok || !c.outs.empty() || (int)pr.s2.size() < c.Cin || pr.out < 0) continue;
        std::vector<float> b9((size_t)9 * c.Cout);
        for (int co = 0; co < c.Cout; ++co) {
            double tap_shift[9];
            for (int t = 0; t < 9; ++t) {
                double acc = 0;
                for (int ci = 0; ci < c.Cin; ++ci) acc += (double)c.weight[((size_t)co * 9 + t) * c.Cin + ci] * pr.t2[ci];
                tap_shift[t] = acc;
            }
            for (int ry = 0; ry < 3; ++ry)
                for (int rx = 0; rx < 3; ++rx) {
                    double b = c.bias[co];
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const bool in_y = !(ry == 0 && ky == 0) && !(ry == 2 && ky == 2);
                            const bool in_x = !(rx == 0 && kx == 0) && !(rx == 2 && kx == 2);
                            if (in_y && in_x) b += tap_shift[ky * 3 + kx];
                        }
                    b9[(size_t)(ry * 3 + rx) * c.Cout + co] = (float)b;
                }
            for (int t = 0; t < 9; ++t)
                for (int ci = 0; ci < c.Cin; ++ci) c.weight[((size_t)co * 9 + t) * c.Cin + ci] *= pr.s2[ci];
        }
        c.bias = b9;                                            // [9][Cout]
        dev_[i].bn_fold_src = c.bn_src;
        dev_[c.bn_src].bn_fold_dst = true;
    }
    for (size_t i = 0; i < plan_.ops.size(); ++i) {
        const POp& op = plan_.ops[i];
        DevOp& d = dev_[i];
        if (op.kind == OpKind::DWPW) {
            d.dww = push(op.dw_weight.data(), op.dw_weight.size());
            d.dwb = push(op.dw_bias.data(), op.dw_bias.size());
        }
        if (op.kind == OpKind::CONV || op.kind == OpKind::GEMM || op.kind == OpKind::DWPW) {
            const int Ktot = op.ks * op.ks * op.Cin;
            d.Kpad = conv_kpad(Ktot);
            const int rows = conv_wt_rows(op.Cout);
            std::vector<float> packed((size_t)rows * d.Kpad, 0.f);
            conv_pack_weights(op.weight.data(), op.Cout, op.Cin, op.ks, packed.data());
            d.wt = push(packed.data(), packed.size());
            // thin 3x3 stride-1 convolutions on large maps (SCRFD's FPN / head convs): weights in MFMA fragment order for conv_halo.hip
            if (op.kind == OpKind::CONV) {
                ConvArgs probe{};
                probe.ks = op.ks; probe.stride = op.stride; probe.pad = op.pad; probe.Cin = op.Cin; probe.Cout = op.Cout; probe.act = (int)op.act;
                probe.res_mode = (int)op.res_mode; probe.H = op.H; probe.W = op.W;
                float dummy = 0.f;
                probe.out2 = op.out2 >= 0 ? &dummy : nullptr;
                probe.bias_cls = dev_[i].bn_fold_src >= 0 ? 1 : 0;   // (a BatchNorm-folded conv carries 9 bias classes: not a halo candidate)
                if (conv_halo_ok(probe) && op.Ho * op.Wo >= 400) {
                    std::vector<float> wf(conv_halo_wfrag_floats(op.Cin, op.Cout));
                    conv_halo_pack_weights(op.weight.data(), op.Cout, op.Cin, wf.data());
                    d.wfrag = push(wf.data(), wf.size());
                    d.halo = true;
                }
            }
            // fused Winograd F(2x2,3x3) image (conv_wino2.hip) for the 3x3 stride-1 convolutions below the F(4x4) threshold: the weights
            // here already carry the block's folded BatchNorm (loop above), whose 9 bias classes the kernel's epilogue applies
            if (op.kind == OpKind::CONV && op.Cin < kWinoMinCin) {           // (merged sibling convolutions too: SCRFD's 64 -> 30 head convolutions)
                ConvArgs probe{};
                probe.ks = op.ks; probe.stride = op.stride; probe.pad = op.pad; probe.Cin = op.Cin; probe.Cout = op.Cout; probe.act = (int)op.act;
                probe.res_mode = (int)op.res_mode; probe.H = op.H; probe.W = op.W; probe.Ho = op.Ho; probe.Wo = op.Wo;
                probe.n_outs = (int)op.outs.size(); probe.bias_cls = dev_[i].bn_fold_src >= 0 ? 1 : 0;
                float dummy2 = 0.f;
                probe.out2 = op.out2 >= 0 ? &dummy2 : nullptr;
                if (wino2_ok(probe)) {
                    std::vector<float> u(wino2_weight_floats(op.Cin, op.Cout));
                    wino2_pack_weights(op.weight.data(), op.Cout, op.Cin, u.data());
                    d.w2 = push(u.data(), u.size());
                    d.w2ok = true;
                }
            }
            // Winograd F(4x4,3x3) image of the same filter: U[f] = G g G^T in fp64, one packed [rows][Cin] matrix per frequency
            if (op.kind == OpKind::CONV && op.ks == 3 && op.stride == 1 && op.pad == 1 && op.Cin >= kWinoMinCin && op.Cin % 32 == 0 &&
                op.Cout % 4 == 0 && op.outs.empty() && op.res_mode != ResMode::UP2X) {
                std::vector<float> u36((size_t)36 * rows * op.Cin, 0.f), uf((size_t)op.Cout * op.Cin);
                std::vector<double> uall((size_t)op.Cout * op.Cin * 36);
                for (int co = 0; co < op.Cout; ++co)
                    for (int ci = 0; ci < op.Cin; ++ci) {
                        double g[9];
                        for (int t = 0; t < 9; ++t) g[t] = op.weight[((size_t)co * 9 + t) * op.Cin + ci];
                        wino_filter_transform(g, &uall[((size_t)co * op.Cin + ci) * 36]);
                    }
                for (int f = 0; f < 36; ++f) {
                    for (size_t e = 0; e < uf.size(); ++e) uf[e] = (float)uall[e * 36 + f];
                    conv_pack_weights(uf.data(), op.Cout, op.Cin, 1, u36.data() + (size_t)f * rows * op.Cin);
                }
                d.w36 = push(u36.data(), u36.size());
                d.w36n = u36.size();
                d.wino = true;
                d.bf2 = wino_gemm_ok_bf16x2(op.Cin, op.Cout);
                const size_t tiles = (size_t)((op.H + 3) / 4) * ((op.W + 3) / 4);
                wino_elems_ = std::max(wino_elems_, 36 * tiles * (size_t)std::max(op.Cin, op.Cout));
                wino_maxc_ = std::max(wino_maxc_, (size_t)std::max(op.Cin, op.Cout));
            }
        } else if (op.kind == OpKind::DWCONV || op.kind == OpKind::DWGLOBAL || op.kind == OpKind::GCONV) {
            d.wt = push(op.weight.data(), op.weight.size());
        }
        if (!op.bias.empty()) d.bias = push(op.bias.data(), op.bias.size());
        if (!op.slope.empty()) { d.slope = push(op.slope.data(), op.slope.size()); d.has_slope = true; }
        if (!op.s2.empty()) { d.s2 = push(op.s2.data(), op.s2.size()); d.t2 = push(op.t2.data(), op.t2.size()); d.has_aff = true; }
    }
    // A strided IResNet block: the 1x1 shortcut runs inside the K loop of the 3x3 convolution it is added to (POp::sc_src)
    for (size_t i = 0; i < plan_.ops.size(); ++i) {
        const POp& c = plan_.ops[i];
        if (c.sc_src < 0 || dev_[i].bn_fold_src >= 0 || c.bias.size() != (size_t)c.Cout) continue;
        // the tenth tap only exists in the direct kernel (conv_mfma.hip): a consumer that may run in its Winograd or halo form
        // (stride-1 projection blocks) keeps the two-convolution form, or the shortcut would silently vanish at large batches
        if (dev_[i].wino || dev_[i].halo) continue;
        const POp& x = plan_.ops[c.sc_src];
        const int K9 = 9 * c.Cin, K = K9 + x.Cin, rows = conv_wt_rows(c.Cout);
        std::vector<float> w((size_t)rows * K, 0.f), b((size_t)c.Cout);
        for (int co = 0; co < c.Cout; ++co) {
            std::copy_n(&c.weight[(size_t)co * K9], K9, &w[(size_t)co * K]);
            std::copy_n(&x.weight[(size_t)co * x.Cin], x.Cin, &w[(size_t)co * K + K9]);
            b[co] = c.bias[co] + (x.bias.empty() ? 0.f : x.bias[co]);
        }
        dev_[i].wt_sc = push(w.data(), w.size());
        dev_[i].bias_sc = push(b.data(), b.size());
        dev_[i].Kpad_sc = K;
        dev_[i].sc_src = c.sc_src;
        dev_[c.sc_src].sc_dst = true;
    }
    // A Winograd conv that is the only reader of its producer's BatchNorm'ed second output takes the producer's plain output
    // instead and lets its input transform apply the affine (exact: padding stays zero): the second output is never written.
    // The planner finds these pairs (POp::bn_src) and keeps the plain output alive up to the consumer.
    for (size_t i = 0; i < plan_.ops.size(); ++i) {
        const int prod = plan_.ops[i].bn_src;
        if (!dev_[i].wino || prod <= 0 || !dev_[prod].has_aff) continue;        // (op 0 = the stem keeps its own path)
        dev_[i].aff_src = prod; dev_[prod].aff_dst = (int)i;
    }
    // Consecutive Winograd convolutions on one small map: the output transform of the first and the input transform of the second
    // run as ONE kernel (winograd.hip, wino_fused_kernel) — the activation between them never makes a round trip through memory.
    for (size_t i = 0; i + 1 < plan_.ops.size(); ++i) {
        const POp& a = plan_.ops[i];
        const POp& b = plan_.ops[i + 1];
        if (!dev_[i].wino || !dev_[i + 1].wino || a.H != b.H || a.W != b.W || a.Cout != b.Cin || !wino_can_fuse(a.H, a.W, a.Cout, false)) continue;
        int feed = -1;
        if (dev_[i + 1].aff_src == (int)i) feed = 1;                       // next conv applies this op's BatchNorm (second output never written)
        else if (a.out >= 0 && b.in == a.out) feed = 0;
        else if (a.out2 >= 0 && b.in == a.out2) feed = 1;
        if (feed < 0) continue;
        int other_uses = 0;                                                // does anything besides the next conv's transform read the plain output?
        if (a.out >= 0) {
            for (const POp& o : plan_.ops)
                for (int x : {o.in, o.in2, o.res}) if (x == a.out) ++other_uses;
            for (const auto& o : plan_.outputs) if (o.tensor == a.out) ++other_uses;
            if (b.in == a.out) --other_uses;                               // that read is what the fused kernel replaces
        }
        const bool writes_out2 = a.out2 >= 0 && dev_[i + 1].aff_src != (int)i;
        if (!wino_can_fuse(a.H, a.W, a.Cout, other_uses > 0 || a.res >= 0 || writes_out2)) continue;
        dev_[i].fuse_next = true; dev_[i].fuse_feed_aff = feed == 1; dev_[i].fuse_keep_out1 = other_uses > 0;
    }
    {   // can the first conv take the u8 image directly?  (3x3, Cin = 3 stored as 4, plain epilogue)
        const POp& op = plan_.ops[0];
        stem_ok_ = op.kind == OpKind::CONV && op.in == plan_.input && op.ks == 3 && op.Cin == 4 && op.res < 0 && op.outs.empty() &&
                   op.Cout % 4 == 0 && op.Cout <= 64 && (op.stride == 1 || op.stride == 2);
        if (stem_ok_) {
            std::vector<float> w27((size_t)27 * op.Cout);
            for (int co = 0; co < op.Cout; ++co)
                for (int t = 0; t < 9; ++t)
                    for (int ci = 0; ci < 3; ++ci) w27[(size_t)(t * 3 + ci) * op.Cout + co] = op.weight[((size_t)co * 9 + t) * 4 + ci];
            dev_[0].w27 = push(w27.data(), w27.size());
            // the same filter with (v - 127.5) / 128 folded in, in the byte order of a BGR pixel (thread-per-pixel stem kernel):
            // w / 128 is exact (power of two); the constant part goes to the bias in fp64
            std::vector<float> wf((size_t)27 * op.Cout), bf((size_t)op.Cout);
            for (int co = 0; co < op.Cout; ++co) {
                double sum = 0;
                for (int t = 0; t < 9; ++t)
                    for (int j = 0; j < 3; ++j) {
                        const float w = op.weight[((size_t)co * 9 + t) * 4 + (2 - j)];
                        wf[(size_t)(t * 3 + j) * op.Cout + co] = w / 128.0f;
                        sum += w;
                    }
                bf[co] = (float)((double)op.bias[co] - 127.5 / 128.0 * sum);
            }
            dev_[0].wf = push(wf.data(), wf.size());
            dev_[0].bf = push(bf.data(), bf.size());
            if (op.Cout % 16 == 0) {                               // matrix-core stems (fused front, stem_mfma_kernel): the same weights as bf16 fragments
                std::vector<unsigned> fr((size_t)(op.Cout / 16) * 3 * 64 * 4, 0u);
                stem_pack_wfrag(wf.data(), op.Cout, fr.data());
                std::vector<float> asf(fr.size());
                memcpy(asf.data(), fr.data(), fr.size() * 4);
                dev_[0].wfr = push(asf.data(), asf.size());
            }
        }
    }
    if (stem_ok_ && plan_.ops.size() > 1) {   // can the stem also be pulled into the depthwise -> pointwise block that consumes it?
        const POp& st = plan_.ops[0];
        const POp& nx = plan_.ops[1];
        int uses = 0;
        for (const auto& o : plan_.ops) for (int x : {o.in, o.in2, o.res}) if (x == st.out) ++uses;
        for (const auto& o : plan_.outputs) if (o.tensor == st.out) ++uses;
        front_ok_ = st.Cout == 16 && st.out >= 0 && st.out2 < 0 && (st.act == Act::NONE || st.act == Act::RELU) && nx.kind == OpKind::DWPW &&
                    nx.in == st.out && uses == 1 && front_fused_ok(nx.Cin, nx.Cout, nx.dw_stride);
    }
    params_.ensure(std::max<size_t>(host.size(), 64) * sizeof(float));
    FH_HIP(hipMemcpy(params_.p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    // host copies of the weights are no longer needed
    for (auto& op : plan_.ops) { std::vector<float>().swap(op.weight); }
}

Net::SkRecord::~SkRecord() {
    if (!p) return;
    // no launch of this Net can write the record once it is back on the free list: drain the NET's device, which need not be the
    // calling thread's current one (one handle per device, destroyed from any thread)
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (device >= 0 && device != cur) (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    if (device >= 0 && device != cur && cur >= 0) (void)hipSetDevice(cur);
    conv_error_record_release(p);
}

void Net::reserve(int max_batch) {
    if (max_batch <= cap_) return;
    cap_ = max_batch;
    arena_.ensure(plan_.arena_elems * (size_t)cap_ * sizeof(float));
    if (partial_.bytes < conv_slab_floats() * sizeof(float)) {
        partial_.ensure(conv_slab_floats() * sizeof(float));
        conv_workspace_init(partial_.as<float>());
        if (!sk_rec_.p) { sk_rec_.p = conv_error_record_new(); (void)hipGetDevice(&sk_rec_.device); }
        if (!sk_rec_.p) throw std::runtime_error("HIP error: cannot allocate the stream-K watchdog record");
        sk_gen_ = conv_error_generation();
    }
    if (wino_elems_) {
        const size_t pad = 36 * 256 * wino_maxc_;                // every frequency plane is padded to whole 256-row tiles
        wino_v_.ensure((wino_elems_ * (size_t)cap_ + pad) * sizeof(float));
        wino_m_.ensure((wino_elems_ * (size_t)cap_ + pad) * sizeof(float));
    }
    // the 4th input lane and alignment gaps must never hold NaNs
    FH_HIP(hipMemset(arena_.p, 0, arena_.bytes));
}

void Net::run_u8(const uint8_t* src, long img_stride, int srcH, int srcW, int step, int batch, hipStream_t s) {
    if (batch <= 0) return;
    if (batch > cap_) throw std::runtime_error("Net::run_u8: batch exceeds reserved capacity");
    // (srcW >= 2: front_kernel reads a row-end dword from the row's last four bytes — a one-pixel-wide image has only three)
    if (stem_ok_ && fuse_stem && front_ok_ && fuse_front && srcW >= 2) {
        // stem conv -> depthwise 3x3 -> pointwise 1x1 in ONE kernel: the stem's 16-channel map (the largest tensor of SCRFD) is never written
        const POp& st = plan_.ops[0];
        const POp& op = plan_.ops[1];
        const DevOp& d0 = dev_[0];
        const DevOp& d = dev_[1];
        const float* P = params_.as<float>();
        ConvArgs a{};
        a.wt = P + d.wt; a.bias = P + d.bias;
        a.out1 = tensor_ptr(op.out);
        a.dw_w = P + d.dww; a.dw_b = P + d.dwb; a.dw_act = (int)op.dw_act; a.dw_stride = op.dw_stride;
        a.B = batch; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout;
        a.ks = 1; a.stride = 1; a.pad = 0; a.Kpad = d.Kpad; a.act = (int)op.act;
        a.u8_src = src; a.u8_img_stride = img_stride; a.u8_step = step; a.u8_srcH = srcH; a.u8_srcW = srcW; a.u8_inH = plan_.inH; a.u8_inW = plan_.inW;
        a.slabs = partial_.as<float>();                                      // (diagnostic builds park their phase stamps there)
        a.u8_stride = st.stride; a.stem_act = (int)st.act; a.stem_wf = P + d0.wf; a.stem_bf = P + d0.bf;
        a.stem_wfrag = reinterpret_cast<const unsigned*>(P + d0.wfr);
        a.cus = cus;
        a.t_flops = 2.0 * (st.macs + op.macs) * batch;
        a.t_bytes = ((double)srcH * srcW * 3 + (double)op.Ho * op.Wo * op.Cout * 4) * batch;        // u8 frame in, pointwise map out
        launch_dwpw(a, s);
        run(batch, s, 2);
        return;
    }
    if (stem_ok_ && fuse_stem) {
        const POp& op = plan_.ops[0];
        const DevOp& d = dev_[0];
        const float* P = params_.as<float>();
        KernelTimer& timer = KernelTimer::get();
        timer.begin(s);
        launch_stem_conv_u8(src, img_stride, srcH, srcW, step, batch, plan_.inH, plan_.inW, op.stride, op.Cout, P + d.w27, P + d.bias, P + d.wf,
                            P + d.bf, d.has_slope ? P + d.slope : nullptr, (int)op.act, op.out >= 0 ? tensor_ptr(op.out) : nullptr,
                            op.out2 >= 0 && !d.bn_fold_dst ? tensor_ptr(op.out2) : nullptr, d.has_aff ? P + d.s2 : nullptr,
                            d.has_aff ? P + d.t2 : nullptr, s, d.wfr ? reinterpret_cast<const unsigned*>(P + d.wfr) : nullptr);
        timer.end(s, 5, 2.0 * op.macs * batch, op.bytes * batch);
        run(batch, s, 1);
        return;
    }
    launch_det_preprocess(src, img_stride, srcH, srcW, step, batch, plan_.inH, plan_.inW, srcH, srcW, input(), s);
    run(batch, s, 0);
}

int Net::set_bf16x2(bool on, hipStream_t s) {
    int layers = 0;
    size_t total = 0;
    for (auto& d : dev_) if (d.wino && d.bf2) { d.w36p = total; total += d.w36n; ++layers; }
    if (on && layers > 0 && w36_bf_.bytes < total * sizeof(float)) {
        w36_bf_.ensure(total * sizeof(float));
        for (const auto& d : dev_)
            if (d.wino && d.bf2) launch_pack_bf16x2(params_.as<float>() + d.w36, w36_bf_.as<float>() + d.w36p, (long)d.w36n, s);
        FH_HIP(hipStreamSynchronize(s));
    }
    bf16x2_ = on && layers > 0;
    return on ? layers : 0;
}

void Net::run(int batch, hipStream_t s, int first_op) {
    if (batch <= 0) return;
    if ((sk_gen_ != conv_error_generation() || conv_error_pending(sk_rec_.p)) && partial_.p) {   // a stream-K hand-off timed out somewhere since: late helper arrivals may have
        conv_workspace_reset_async(partial_.as<float>(), s);  // left counters non-zero — re-zero them, stream-ordered, before the next launch
        sk_gen_ = conv_error_generation();
    }
    if (batch > cap_) throw std::runtime_error("Net::run: batch exceeds reserved capacity");
    const float* P = params_.as<float>();
    KernelTimer& timer = KernelTimer::get();
    bool v_ready = false;                                 // the Winograd V workspace already holds the NEXT op's input transform
    for (size_t i = (size_t)first_op; i < plan_.ops.size(); ++i) {
        const POp& op = plan_.ops[i];
        const DevOp& d = dev_[i];
        int tag = 5;
        const bool dense = op.kind == OpKind::CONV || op.kind == OpKind::GEMM || op.kind == OpKind::DWPW;   // the conv launcher times its own kernels
        if (!dense) timer.begin(s);
        switch (op.kind) {
            case OpKind::CONV:
            case OpKind::GEMM: {
                if (d.sc_dst && fold_shortcut && force_cfg < 0) break;   // runs inside its consumer's K loop
                ConvArgs a{};
                a.in = tensor_ptr(op.in);
                a.wt = P + d.wt;
                a.bias = P + d.bias;
                a.slope = d.has_slope ? P + d.slope : nullptr;
                a.res = op.res >= 0 ? tensor_ptr(op.res) : nullptr;
                a.out1 = op.out >= 0 ? tensor_ptr(op.out) : nullptr;
                a.out2 = op.out2 >= 0 ? tensor_ptr(op.out2) : nullptr;
                if (d.bn_fold_dst) a.out2 = nullptr;                       // its only reader has the BatchNorm folded into its weights
                if (d.bn_fold_src >= 0) { a.in = tensor_ptr(plan_.ops[d.bn_fold_src].out); a.bias_cls = 1; }
                if (d.aff_dst >= 0 && winograd) {                          // its only reader applies the BatchNorm itself (see aff_src)
                    const POp& c = plan_.ops[d.aff_dst];
                    if ((long)batch * ((c.H + 3) / 4) * ((c.W + 3) / 4) >= kWinoMinTiles) a.out2 = nullptr;
                }
                a.s2 = d.has_aff ? P + d.s2 : nullptr;
                a.t2 = d.has_aff ? P + d.t2 : nullptr;
                a.slabs = partial_.as<float>();
                a.sk_err = sk_rec_.p;
                a.sk_enable = sk_enable ? 1 : 0;
                a.cus = cus;
                a.no_pw = force_cfg >= 0 ? 1 : 0;
                a.B = batch; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout;
                a.ks = op.ks; a.stride = op.stride; a.pad = op.pad; a.Kpad = d.Kpad;
                a.act = (int)op.act; a.res_mode = (int)op.res_mode;
                a.n_outs = (int)op.outs.size();
                for (int g = 0; g < a.n_outs && g < 3; ++g) { a.outs[g] = tensor_ptr(op.outs[g]); a.oact[g] = op.out_act[g]; a.oc0[g] = op.out_c0[g]; }
                if (a.n_outs > 0) a.oc0[a.n_outs] = op.out_c0[a.n_outs];
                const long M = (long)batch * op.Ho * op.Wo;
                int cfg = force_cfg >= 0 ? force_cfg : conv_pick_cfg(M, op.Cout);
                if (force_cfg < 0 && cfg == 0 && ((M + 127) / 128) * ((op.Cout + 127) / 128) < 128) cfg = 3;   // very few tiles: go finer
                a.t_flops = 2.0 * op.macs * batch; a.t_bytes = op.bytes * batch;
                if (d.sc_src >= 0 && fold_shortcut && force_cfg < 0) {
                    const POp& x = plan_.ops[d.sc_src];
                    a.wt = P + d.wt_sc; a.bias = P + d.bias_sc; a.Kpad = d.Kpad_sc;
                    a.res = nullptr; a.res_mode = (int)ResMode::NONE;
                    a.sc_in = tensor_ptr(x.in); a.sc_H = x.H; a.sc_W = x.W; a.sc_C = x.Cin; a.sc_stride = x.stride;
                    a.t_flops += 2.0 * x.macs * batch;
                }
                // (small batches: the 36 GEMMs would be mostly tile padding and the direct form with split-K is faster —
                //  measured cross-over at 256 tiles per GEMM: B = 1: 0.98 ms direct / 1.83 ms Winograd, B = 32: 4.82 / 3.88)
                if (d.halo && halo_conv && force_cfg < 0) {
                    launch_conv_halo(a, P + d.wfrag, s);
                    tag = 9;
                    break;
                }
                // fused F(2x2,3x3) for the 64-channel 3x3 layers once there is about one one-wave workgroup per CU (wino2_blocks counts them
                // in fours); below that the direct kernel's split-K fills the chip better
                if (d.w2ok && winograd && force_cfg < 0 && d.sc_src < 0 && wino2_blocks(a) >= wino2_min_blocks()) {
                    a.wt = P + d.w2;
                    launch_wino2(a, s);
                    break;
                }
                const bool use_wino = d.wino && winograd && (long)batch * ((op.H + 3) / 4) * ((op.W + 3) / 4) >= kWinoMinTiles;
                if (use_wino) {
                    // 36 GEMMs of depth Cin: short K loops, so the 128x32 tile (4 workgroups per CU) beats the 128x128 one
                    // (IResNet-50, B = 128: 12.75 ms against 14.72 ms)
                    const int wcfg = force_cfg >= 0 ? force_cfg : 2;
                    const bool bf2 = bf16x2_ && d.bf2 && force_cfg < 0;
                    // maps of side 4 k + 2 (14x14): the mixed F(4x4) / F(2x2) tiling (winograd.hip, WinoPlanes) — its own V / M layout, so
                    // the producer of V and the consumer must agree: the decision depends on the map and the batch only
                    WinoPlanes pl;
                    const bool mix = wcfg == 2 && wino_mix_layout(batch, op.H, op.W, op.Cin, op.Cout, &pl);
                    if (!v_ready) {                                     // (else the previous layer's fused transform already wrote V)
                        const float* in_s = nullptr; const float* in_t = nullptr;
                        if (d.aff_src >= 0) {
                            a.in = tensor_ptr(plan_.ops[d.aff_src].out);
                            in_s = P + dev_[d.aff_src].s2; in_t = P + dev_[d.aff_src].t2;
                        }
                        if (mix) launch_wino_mix(a, pl, nullptr, wino_v_.as<float>(), 0, in_s, in_t, bf2, s);
                        else launch_wino_input(a, wino_v_.as<float>(), in_s, in_t, bf2, s);
                    }
                    launch_wino_gemm(a, bf2 ? w36_bf_.as<float>() + d.w36p : P + d.w36, wino_v_.as<float>(), wino_m_.as<float>(), wcfg, bf2, s,
                                     mix ? &pl : nullptr);
                    v_ready = false;
                    if (d.fuse_next && fuse_wino && i + 1 < plan_.ops.size()) {
                        const POp& nx = plan_.ops[i + 1];
                        const bool next_wino = dev_[i + 1].wino && (long)batch * ((nx.H + 3) / 4) * ((nx.W + 3) / 4) >= kWinoMinTiles;
                        const bool next_mix = next_wino && wcfg == 2 && wino_mix_layout(batch, nx.H, nx.W, nx.Cin, nx.Cout, nullptr);
                        if (next_wino && mix == next_mix) {
                            ConvArgs e = a;
                            if (!d.fuse_keep_out1) e.out1 = nullptr;
                            const bool pack_next = bf16x2_ && dev_[i + 1].bf2 && force_cfg < 0;
                            if (mix) launch_wino_mix(e, pl, wino_m_.as<float>(), wino_v_.as<float>(), d.fuse_feed_aff ? 1 : 0, nullptr, nullptr, pack_next, s);
                            else launch_wino_fused(e, wino_m_.as<float>(), wino_v_.as<float>(), d.fuse_feed_aff ? 1 : 0, pack_next, s);
                            v_ready = true;
                        }
                    }
                    if (!v_ready) {
                        if (mix) launch_wino_mix(a, pl, wino_m_.as<float>(), nullptr, 0, nullptr, nullptr, false, s);
                        else launch_wino_output(a, wino_m_.as<float>(), s);
                    }
                    tag = 7;
                    break;
                }
                launch_conv(a, cfg, s);
                tag = cfg;
                break;
            }
            case OpKind::DWPW: {
                ConvArgs a{};
                a.in = tensor_ptr(op.in);
                a.wt = P + d.wt; a.bias = P + d.bias;
                a.out1 = tensor_ptr(op.out);
                a.dw_w = P + d.dww; a.dw_b = P + d.dwb; a.dw_act = (int)op.dw_act; a.dw_stride = op.dw_stride;
                a.B = batch; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout;
                a.ks = 1; a.stride = 1; a.pad = 0; a.Kpad = d.Kpad; a.act = (int)op.act;
                a.t_flops = 2.0 * op.macs * batch; a.t_bytes = op.bytes * batch;
                a.slabs = partial_.as<float>();                             // (diagnostic builds park their phase stamps there)
                launch_dwpw(a, s);
                break;
            }
            case OpKind::DWCONV:
                launch_dwconv3x3(tensor_ptr(op.in), P + d.wt, P + d.bias, tensor_ptr(op.out), batch, op.H, op.W, op.Cin, op.stride,
                                 (int)op.act, d.has_slope ? P + d.slope : nullptr, s);
                tag = 4;
                break;
            case OpKind::DWGLOBAL:
                launch_dwglobal(tensor_ptr(op.in), P + d.wt, P + d.bias, tensor_ptr(op.out), batch, op.ks * op.ks, op.Cin, (int)op.act,
                                d.has_slope ? P + d.slope : nullptr, s);
                break;
            case OpKind::GCONV:
                launch_gconv3x3(tensor_ptr(op.in), P + d.wt, P + d.bias, tensor_ptr(op.out), batch, op.H, op.W, op.Cin,
                                (int)(op.weight_group), op.stride, (int)op.act, d.has_slope ? P + d.slope : nullptr, s);
                tag = 4;
                break;
            case OpKind::AFFINE:
                launch_affine(tensor_ptr(op.in), P + d.s2, P + d.t2, tensor_ptr(op.out), (long)batch * op.H * op.W, op.Cin, s);
                break;
            case OpKind::ACT:
                launch_act(tensor_ptr(op.in), d.has_slope ? P + d.slope : nullptr, tensor_ptr(op.out), (long)batch * op.H * op.W,
                           op.Cin, (int)op.act, s);
                break;
            case OpKind::ADD:
                launch_add(tensor_ptr(op.in), tensor_ptr(op.in2), tensor_ptr(op.out), (long)batch * op.H * op.W * op.Cin, s);
                break;
            case OpKind::UPSAMPLE:
                launch_upsample2x(tensor_ptr(op.in), tensor_ptr(op.out), batch, op.H, op.W, op.Cin, s);
                break;
        }
        if (!dense) timer.end(s, tag, 2.0 * op.macs * batch, op.bytes * batch);
        if (debug_sync()) {                                   // FACEHIP_DEBUG_SYNC=1: localise a failing launch to its op
            fprintf(stderr, "[facehip] op %zu %s batch %d done-launch\n", i, op.name.c_str(), batch); fflush(stderr);
            const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(s);
            if (e1 != hipSuccess || e2 != hipSuccess)
                throw std::runtime_error("HIP error after op " + std::to_string(i) + " (" + op.name + "): " + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        }
    }
    FH_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ Detector
Detector::Detector(const std::string& onnx_path) : net_(onnx_path, 640, 640) {   // defaults: src/face_detector.cpp:8-9
    const auto& outs = net_.plan().outputs;
    const int H = net_.in_h(), W = net_.in_w();
    if (outs.size() == 9) {
        static const int cols[9] = {1, 1, 1, 4, 4, 4, 10, 10, 10};
        static const int strides[3] = {8, 16, 32};
        bool ok = true;
        for (int i = 0; i < 9; ++i) {
            const int s = strides[i % 3];
            ok = ok && outs[i].cols == cols[i] && outs[i].rows == (H / s) * (W / s) * 2;
        }
        if (ok) anchors_ = ((H / 8) * (W / 8) + (H / 16) * (W / 16) + (H / 32) * (W / 32)) * 2;
    } else if (outs.size() >= 1 && outs[0].cols >= 15) {
        // the reference's own assumption: output 0 already is [*, N, >=15] (src/face_detector.cpp:242-325)
        predecoded_ = true;
        anchors_ = outs[0].rows;
        feat_ = outs[0].cols;
    }
    // anything else: "Unexpected output shape format" -> zero faces (src/face_detector.cpp:326-328)
    cap_ = 1;
    while (cap_ < std::max(anchors_, 1)) cap_ <<= 1;
}

void Detector::reserve(int n, int rows, int cols) {
    (void)rows; (void)cols;
    net_.reserve(n);
    if (n > nb_) {
        nb_ = n;
        cand_.ensure((size_t)n * cap_ * sizeof(FaceRec));
        keys_.ensure((size_t)n * cap_ * sizeof(unsigned long long));
        ws_.ensure((size_t)n * cap_ * sizeof(int));
        count_.ensure((size_t)n * sizeof(int));
    }
}

void Detector::run_network_dev(const uint8_t* frames, int n, int rows, int cols, int step, long stride, hipStream_t s) {
    reserve(n, rows, cols);
    const int inW = net_.in_w(), inH = net_.in_h();
    // src/face_detector.cpp:101-106
    const float scaleW = (float)inW / (float)cols, scaleH = (float)inH / (float)rows;
    scale_ = std::min(scaleW, scaleH);
    const int newW = (int)((float)cols * scale_), newH = (int)((float)rows * scale_);
    if (newW <= 0 || newH <= 0) throw std::runtime_error("Invalid resize dimensions");   // :109-113
    const uint8_t* src = frames;
    long sstride = stride;
    int sstep = step;
    if (newW != cols || newH != rows) {
        resized_.ensure((size_t)n * newH * newW * 3);
        launch_resize_u8c3(frames, stride, rows, cols, step, resized_.as<uint8_t>(), (long)newH * newW * 3, newH, newW, newW * 3, n, s);
        src = resized_.as<uint8_t>(); sstride = (long)newH * newW * 3; sstep = newW * 3;
    }
    net_.run_u8(src, sstride, newH, newW, sstep, n, s);
}

void Detector::postprocess_dev(int n, float score_thr, float nms_thr, FaceRec* out, int max_out, int* counts, hipStream_t s) {
    if (anchors_ <= 0) { FH_HIP(hipMemsetAsync(counts, 0, (size_t)n * sizeof(int), s)); return; }
    FH_HIP(hipMemsetAsync(count_.p, 0, (size_t)n * sizeof(int), s));
    if (predecoded_) {
        launch_rows_threshold(net_.output(0), n, anchors_, feat_, scale_, score_thr, cand_.as<FaceRec>(),
                              keys_.as<unsigned long long>(), count_.as<int>(), cap_, s);
    } else {
        DecodeArgs a{};
        for (int i = 0; i < 3; ++i) { a.score[i] = net_.output(i); a.bbox[i] = net_.output(3 + i); a.kps[i] = net_.output(6 + i); }
        a.inH = net_.in_h(); a.inW = net_.in_w(); a.B = n; a.scale = scale_; a.thr = score_thr;
        a.cand = cand_.as<FaceRec>(); a.keys = keys_.as<unsigned long long>(); a.count = count_.as<int>(); a.cap = cap_;
        launch_scrfd_decode(a, s);
    }
    launch_sort_nms(cand_.as<FaceRec>(), keys_.as<unsigned long long>(), count_.as<int>(), cap_, n, nms_thr, out, counts, max_out,
                    ws_.as<int>(), s);
    FH_HIP(hipGetLastError());
}

void Detector::detect_dev(const uint8_t* frames, int n, int rows, int cols, int step, long stride, float score_thr, float nms_thr,
                          FaceRec* out, int max_out, int* counts, hipStream_t s) {
    if (n <= 0) return;
    run_network_dev(frames, n, rows, cols, step, stride, s);
    postprocess_dev(n, score_thr, nms_thr, out, max_out, counts, s);
}

// ------------------------------------------------------------------------------------------ Recognizer
Recognizer::Recognizer(const std::string& onnx_path) : net_(onnx_path, 112, 112) {      // src/face_recognizer.cpp:8-9
    const auto& o = net_.plan().outputs.at(0);
    dim_ = o.rows * o.cols;                       // feature size comes from the output shape (:286-294)
}

void Recognizer::embed_aligned_dev(const uint8_t* crops, int n, float* out, hipStream_t s, float* raw_out) {
    const int H = net_.in_h(), W = net_.in_w();
    net_.reserve(std::min(n, max_chunk));
    for (int off = 0; off < n; off += max_chunk) {
        const int c = std::min(max_chunk, n - off);
        net_.run_u8(crops + (size_t)off * H * W * 3, (long)H * W * 3, H, W, W * 3, c, s);
        if (raw_out) FH_HIP(hipMemcpyAsync(raw_out + (size_t)off * dim_, net_.output(0), (size_t)c * dim_ * sizeof(float), hipMemcpyDeviceToDevice, s));
        launch_l2_normalize(net_.output(0), out + (size_t)off * dim_, c, dim_, s);
    }
    FH_HIP(hipGetLastError());
}

void Recognizer::align_dev(const uint8_t* frames, int rows, int cols, int step, long stride, const FaceRec* faces,
                           const int* frame_of, int n, uint8_t* crops, int* ok, hipStream_t s) {
    launch_align(frames, stride, rows, cols, step, faces, frame_of, n, net_.in_h(), net_.in_w(), crops, ok, s);
    FH_HIP(hipGetLastError());
}

void Recognizer::embed_faces_dev(const uint8_t* frames, int rows, int cols, int step, long stride, const FaceRec* faces,
                                 const int* frame_of, int n, float* out, int* ok, hipStream_t s) {
    if (n <= 0) return;
    const size_t crop = (size_t)net_.in_h() * net_.in_w() * 3;
    crops_.ensure((size_t)n * crop);
    ok_.ensure((size_t)n * sizeof(int));
    int* okp = ok ? ok : ok_.as<int>();
    align_dev(frames, rows, cols, step, stride, faces, frame_of, n, crops_.as<uint8_t>(), okp, s);
    embed_aligned_dev(crops_.as<uint8_t>(), n, out, s);
}

void Recognizer::resize_embed_dev(const uint8_t* frames, int n, int rows, int cols, int step, long stride, float* out, hipStream_t s) {
    // extractFeatureSimple (src/face_recognizer.cpp:152-234): cv::resize the whole image to the input size
    const int H = net_.in_h(), W = net_.in_w();
    crops_.ensure((size_t)n * H * W * 3);
    launch_resize_u8c3(frames, stride, rows, cols, step, crops_.as<uint8_t>(), (long)H * W * 3, H, W, W * 3, n, s);
    embed_aligned_dev(crops_.as<uint8_t>(), n, out, s);
}

// ------------------------------------------------------------------------------------------ Gallery
void Gallery::upload(const float* rows, long n, bool device_src, long index_base) {
    if (dim_ % 64) throw std::runtime_error("gallery: dim must be a multiple of 64");
    rows_.ensure((size_t)n * dim_ * sizeof(float));
    FH_HIP(hipMemcpy(rows_.p, rows, (size_t)n * dim_ * sizeof(float), device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    n_ = n; base_ = index_base;
}

long Gallery::enroll(const float* rows, long n, bool device_src) {
    if (dim_ % 64) throw std::runtime_error("gallery: dim must be a multiple of 64");
    const size_t row_bytes = (size_t)dim_ * sizeof(float);
    const size_t need = (size_t)(n_ + n) * row_bytes;
    if (need > rows_.bytes) {                                    // grow geometrically, keeping the enrolled rows
        DevBuf bigger;
        bigger.ensure(std::max(need, rows_.bytes * 2));
        if (n_ > 0) FH_HIP(hipMemcpy(bigger.p, rows_.p, (size_t)n_ * row_bytes, hipMemcpyDeviceToDevice));
        std::swap(bigger.p, rows_.p);
        std::swap(bigger.bytes, rows_.bytes);
    }
    FH_HIP(hipMemcpy(static_cast<char*>(rows_.p) + (size_t)n_ * row_bytes, rows, (size_t)n * row_bytes,
                     device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    const long first = base_ + n_;
    n_ += n;
    return first;
}

void Gallery::label_dev(const float* q, int Q, float thr, int* out_label, float* out_score, hipStream_t s) {
    best_i_.ensure((size_t)std::max(Q, 1) * sizeof(int));
    topk_dev(q, Q, 1, out_score, best_i_.as<int>(), s);
    launch_label(out_score, best_i_.as<int>(), Q, thr, out_label, s);
}

void Gallery::topk_dev(const float* q, int Q, int k, float* out_score, int* out_idx, hipStream_t s) {
    if (Q <= 0 || Q > 256 || k <= 0 || k > 16) throw std::runtime_error("gallery: need 0 < Q <= 256 and 0 < k <= 16");
    if (dim_ % 64) throw std::runtime_error("gallery: dim must be a multiple of 64");
    // queries as the GEMM's N operand: whole 64-row tiles, zero rows behind Q (only that tail is cleared)
    const int qrows = (Q + 63) / 64 * 64;
    qpack_.ensure((size_t)qrows * dim_ * sizeof(float));
    if (qrows > Q) FH_HIP(hipMemsetAsync(qpack_.as<float>() + (size_t)Q * dim_, 0, (size_t)(qrows - Q) * dim_ * sizeof(float), s));
    FH_HIP(hipMemcpyAsync(qpack_.p, q, (size_t)Q * dim_ * sizeof(float), hipMemcpyDeviceToDevice, s));
    int tpp = 0;
    const int parts = n_ > 0 ? gallery_parts(n_, Q, &tpp) : 0;
    ps_.ensure((size_t)std::max(parts, 1) * Q * k * sizeof(float));
    pi_.ensure((size_t)std::max(parts, 1) * Q * k * sizeof(int));
    // ONE pass over the gallery: dot products stay in the MFMA accumulators, per-workgroup top-k lists come out (gallery.hip)
    seed_s_.ensure((size_t)Q * k * sizeof(float));
    seed_i_.ensure((size_t)Q * k * sizeof(int));
    launch_gallery_topk(rows_.as<float>(), n_, dim_, qpack_.as<float>(), Q, k, base_, ps_.as<float>(), pi_.as<int>(), seed_s_.as<float>(),
                        seed_i_.as<int>(), s);
    launch_topk_merge(ps_.as<float>(), pi_.as<int>(), parts, Q, k, out_score, out_idx, s);
    FH_HIP(hipGetLastError());
}

}  // namespace fh
