// kernels.h — host-side launch interface of the gfx950 kernels (device pointers only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include <vector>

namespace fh {

void hip_check(hipError_t e, const char* what);
#define FH_HIP(x) ::fh::hip_check((x), #x)

// Orders one wave's own LDS stores before its own later LDS loads of bytes OTHER lanes stored (the per-wave turn-around scratch of the
// whole-line epilogues) and those loads before the next round's stores.  LDS operations of one wave execute in issue order, so no
// instruction is needed; this only keeps the COMPILER from moving an access across one it cannot prove disjoint (wavefront-scope fence +
// wave barrier: no code is emitted).
#if defined(__HIPCC__)
__device__ __forceinline__ void wave_lds_order() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}
#endif

// Optional per-launch HIP-event timing of the network kernels (bench.py's roofline leg).
// Tags: 0..3 = conv_igemm tile configs, 4 = depthwise conv, 5 = other graph ops, 6 = conv stream-K fix-up.
struct KernelTimer {
    static constexpr int kTags = 13;     // 0-3 conv tile configs, 4 dw / dwpw, 5 other, 6 fix-up, 7 Winograd GEMM, 8 Winograd transforms, 9 halo conv, 10 conv_tall_kernel, 11 conv_pw_kernel, 12 wino2_kernel
    bool enabled = false;
    void begin(hipStream_t s);
    void end(hipStream_t s, int tag, double flops, double bytes);
    // synchronises, aggregates elapsed ms / flops / bytes / launches per tag, then resets
    void collect(double* ms, double* flops, double* bytes, long long* launches);
    // un-aggregated variant (tuning): one entry per recorded launch, in launch order
    int collect_ops(double* ms, double* flops, int* tag, int cap);
    static KernelTimer& get();

  private:
    struct Rec { hipEvent_t a, b; int tag; double flops, bytes; };
    std::vector<Rec> recs_;
    std::vector<hipEvent_t> pool_;
    hipEvent_t cur_ = nullptr;
    hipEvent_t take();
};


// --------------------------------------------------------------------------------------------
// Dense convolution / FC as implicit GEMM on v_mfma_f32_32x32x2_f32 (conv_mfma.hip)
//   M = B*Ho*Wo output pixels, N = Cout, K = ks*ks*Cin (k = tap*Cin + ci), all fp32, NHWC.
// --------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* in;        // [B,H,W,Cin]
    const float* wt;        // packed [CoutPad][Kpad], rows >= Cout and cols >= Ktot are zero
    const float* bias;      // [Cout] — or, with bias_cls != 0, [9][Cout]: one vector per border class of the output pixel
                            // (3 * (oy == 0 ? 0 : oy == Ho-1 ? 2 : 1) + same for ox): a pre-conv BatchNorm folded into this 3x3 pad-1 conv
    const float* slope;     // [Cout] (PReLU) or null
    const float* res;       // residual [B,Ho,Wo,Cout] (SAME) / [B,Ho/2,Wo/2,Cout] (UP2X) or null
    float* out1;            // [B,Ho,Wo,Cout] or null
    float* out2;            // second output  out*s2 + t2, or null
    const float* s2;
    const float* t2;
    float* slabs;           // stream-K accumulator slabs (conv_slab_floats() floats) or null = plain tiles only
    const float* zeros;     // >= 16 zero bytes (filled in by launch_conv)
    int B, H, W, Cin, Ho, Wo, Cout;
    int ks, stride, pad;
    int Kpad;               // multiple of 32
    int bias_cls;           // see bias
    // folded shortcut (sc_in != null; needs ks == 3, Cin % 32 == 0, sc_C % 32 == 0): a 1x1 stride-sc_stride convolution of sc_in
    // [B,sc_H,sc_W,sc_C] on the same output grid runs as a tenth tap of the K loop — wt holds its sc_C columns after the 9*Cin
    // own ones (Kpad = 9*Cin + sc_C), bias the sum of both biases
    const float* sc_in;
    int sc_H, sc_W, sc_C, sc_stride;
    int act;                // fh::Act
    int res_mode;           // fh::ResMode
    const float* dw_w;      // fused depthwise 3x3 front end (launch_dwpw): weights [9][Cin], bias [Cin], activation
    const float* dw_b;
    int dw_act;
    int dw_stride;          // 1 | 2
    // fused u8 stem in FRONT of the depthwise part (launch_dwpw with u8_src != null): `in` is not read — the workgroup computes the
    // stem convolution (3x3, 3 -> 16 channels, stride u8_stride, preprocess folded into stem_wf / stem_bf as for launch_stem_conv_u8)
    // for its halo pixels straight from the BGR u8 frames.  H x W = the stem's output grid = the depthwise input.
    const uint8_t* u8_src;
    long u8_img_stride;
    int u8_step, u8_srcH, u8_srcW, u8_inH, u8_inW, u8_stride, stem_act;
    const float* stem_wf;
    const float* stem_bf;
    const unsigned* stem_wfrag;   // launch_dwpw's fused stem: stem_wf as bf16 MFMA fragments (stem_pack_wfrag)
    int no_prio;            // 1: the persistent depthwise kernels do not rotate their wave priority (dwpw_mfma.hip rotate_wave_priority: A / B switch)
    int no_pw;              // 1: keep conv_igemm_kernel even where conv_pw_kernel (the lean 1x1 form) would take the layer (forced-cfg runs)
    int n_outs;             // > 0: merged sibling convs — channels [oc0[g], oc0[g+1]) go to outs[g] with act oact[g]
    float* outs[3];
    int oc0[4];
    int oact[3];
    int sk_enable;          // allow the stream-K remainder wave
    int cus;                // CUs this launch can occupy (0 = the whole device); sizes the stream-K remainder round
    // grouped GEMM (Winograd: 36 independent [rows x K] x [K x N] products stacked along M in one launch): rows
    // [g*wt_group_rows, (g+1)*wt_group_rows) use the weight matrix wt + g*wt_gs (floats).  wt_group_rows = 0: off;
    // otherwise a multiple of the tile height (128 covers every configuration but the 256-row one).
    int wt_group_rows;
    long wt_gs;
    double t_flops, t_bytes; // algorithmic work of this launch (only used by the optional KernelTimer)
    // filled in by launch_conv: #plain tiles, K-chunk units dealt to helpers, units per helper, #helpers, #remainder
    // tiles, chunks each owner computes itself (0 = no owners: fix-up kernel), slab slots per remainder tile
    int sk_full, sk_units, sk_q, sk_helpers, sk_rem, sk_owner_chunks, sk_maxp;
    // stream-K watchdog (filled in by launch_conv): host-visible error record, the owners' wait bound in 2^16 ticks of the 100 MHz
    // wall clock, and the test hook that makes helpers "lose" their publication
    unsigned* sk_err;
    int ep_direct;                  // tuning hook (FACEHIP_EP_DIRECT): 1 = epilogue stores straight from the accumulator registers, 3 = whole-line form for every Cout % 4 == 0
    unsigned sk_timeout;
    int sk_test_drop;
    int tile0;              // filled in by launch_conv: first tile conv_igemm_kernel computes (the tiles before it ran in conv_tall_kernel)
};

// cfg: 0 = 128x128 tile, 1 = 256x64, 2 = 128x32, 3 = 64x64 (256 threads each); -1 = choose.
void launch_conv(const ConvArgs& a, int cfg, hipStream_t s);
int conv_pick_cfg(long M, int Cout);
void conv_workspace_init(float* ws);          // zero the counter words of a freshly allocated stream-K workspace
void conv_workspace_reset_async(float* ws, hipStream_t s);   // the same, stream-ordered (after a reported hand-off time-out)
// Stream-K watchdog.  An owner workgroup whose helpers never arrive gives up after a bounded wait and reports through a host-mapped
// record of the launching Net's own; conv_take_error(msg, rec) — called with the records of the handles an API call works on — returns
// true once for that handle (with a message starting "HIP error") and bumps conv_error_generation(), after which
// every Net re-zeroes its hand-off counters before its next run.  conv_debug_streamk(drop, ms): test hook — helpers skip their
// publication / the owners' wait bound in milliseconds (0 = the 2 s default).
unsigned* conv_error_words();                 // the process-wide record (launches whose ConvArgs::sk_err is null: the single-kernel test entry points)
unsigned* conv_error_record_new();            // a record of a Net's own (host-mapped, 64 B); ConvArgs::sk_err of its launches
void conv_error_record_release(unsigned* r);
bool conv_take_error(std::string& msg, unsigned* rec);   // consumes THIS record's report, if any
bool conv_error_pending(const unsigned* rec); // a reported time-out has not been consumed yet
unsigned conv_error_generation();
void conv_debug_streamk(int drop_publish, int timeout_ms);
unsigned conv_debug_generation();             // bumped by conv_debug_streamk (its settings are kernel arguments: captured graphs hold them)
// Fused Winograd F(2x2, 3x3) for 3x3 stride-1 pad-1 convolutions with Cin = 64 and Cout % 64 == 0 (or <= 32 with merged outputs) (conv_wino2.hip): a.wt = the
// wino2_pack_weights image of the filter [Cout][9][Cin]; epilogue fields (bias / bias_cls, act, slope, res, out1, out2) as for launch_conv
size_t wino2_weight_floats(int Cin, int Cout);
void wino2_pack_weights(const float* w_ohwi, int Cout, int Cin, float* dst);
bool wino2_ok(const ConvArgs& a);
void launch_wino2(const ConvArgs& a, hipStream_t s);
long wino2_blocks(const ConvArgs& a);          // workgroups the launch would have
double wino2_debug_clock_mhz();                // diagnostic builds (scripts/wino2_prof.sh): median in-kernel shader clock of the last launch
const float* conv_zero_line();                // 8 KiB of device zeros (target of padded / dead loads)
int conv_num_cus();                           // compute units of the current device
// dense 3x3 stride-1 convolutions with 16 input channels and <= 64 output channels on an 8x16 spatial tile with an LDS halo
// (conv_halo.hip); wfrag = conv_halo_pack_weights' image of the filter
bool conv_halo_ok(const ConvArgs& a);
size_t conv_halo_wfrag_floats(int Cin, int Cout);
void conv_halo_pack_weights(const float* w_ohwi, int Cout, int Cin, float* dst);
void launch_conv_halo(const ConvArgs& a, const float* wfrag, hipStream_t s);
// depthwise 3x3 stride 1 (+bias +act) fused with the 1x1 conv that consumes it (dwpw_mfma.hip)
void stem_pack_wfrag(const float* wf, int Cout, unsigned* out);   // wf [27][Cout] (folded stem weights) -> [Cout/16][3][64][4] dwords, see dwpw_mfma.hip
bool front_fused_ok(int Cin, int Cout, int dw_stride);   // the opening block (u8 stem + depthwise + pointwise) has a one-kernel form
void launch_dwpw(const ConvArgs& a, hipStream_t s);
// Winograd F(4x4,3x3) form of a 3x3 stride-1 pad-1 convolution (winograd.hip): a = the convolution's arguments,
// wt36 = U[36][conv_wt_rows(Cout)][Cin], V / M = workspaces of 36 * tiles * max(Cin, Cout) floats each
// in_scale / in_shift (optional): per-input-channel affine applied to in-image pixels by the input transform
void launch_conv_winograd(const ConvArgs& a, const float* wt36, float* V, float* M, int cfg, const float* in_scale, const float* in_shift,
                          hipStream_t s);
// the three stages separately, and stage 3 of one convolution fused with stage 1 of the next (same map, C = Cout = next Cin)
// pack / bf16x2 / pack_next: the opt-in split-bf16 operand format (V and U as (hi, mid) bf16 pairs in 32-bit words, see winograd.hip)
struct WinoPlanes;
void launch_wino_input(const ConvArgs& a, float* V, const float* in_scale, const float* in_shift, bool pack, hipStream_t s);
void launch_wino_gemm(const ConvArgs& a, const float* wt36, const float* V, float* M, int cfg, bool bf16x2, hipStream_t s,
                      const WinoPlanes* mix = nullptr);   // mix: V / M in the mixed-tiling layout (lean kernel only)
bool wino_gemm_ok_bf16x2(int Cin, int Cout);
void launch_wino_output(const ConvArgs& a, const float* M, hipStream_t s);
bool wino_can_fuse(int H, int W, int C, bool touches_memory);
void launch_wino_fused(const ConvArgs& a, const float* M, float* Vnext, int feed_aff, bool pack_next, hipStream_t s);
void launch_pack_bf16x2(const float* in, float* out, long n, hipStream_t s);      // fp32 -> split-bf16 words, n % 4 == 0
// Mixed F(4x4) / F(2x2) tiling (round 3) for maps whose side is 4 k + 2 or 4 k + 1 (IResNet's 14x14): the last tile row / column is an
// F(2x2, 3x3) tile instead of an F(4x4) tile hanging over the border — 484 instead of 576 GEMM rows per 14x14 image.  Tiles fall into
// four classes (rows F4 | F2) x (columns F4 | F2) with 36 / 24 / 24 / 16 frequency planes; the F(2) interpolation points {0, +-1, inf}
// are a subset of F(4)'s, so the planes multiply the SAME 36 weight matrices (plane (i, j) of an F(2) direction uses frequency
// {0, 1, 2, 5}[i] with the row scales {4, -3, -3, 1} folded into the input transform).
struct WinoPlanes {
    int e[4];                // first 128-row GEMM tile of class c (classes stored one after the other, plane-major inside a class)
    int t[4];                // 128-row GEMM tiles per plane
    int nfc[4];              // frequencies along a patch row: 6 (F4 columns) or 4 (F2 columns)
    int f2[4];               // bit 0: columns are F(2), bit 1: rows are F(2)
    int n[4];                // tiles per image
    int rows[4];             // rows per plane (B * n rounded up to 128)
    int base[4];             // first row of the class's first plane
    int total_tiles;         // 128-row GEMM tiles over all planes
    int planes;              // 36 + 24 + 24 + 16 (classes with n = 0 contribute none)
};
// layout of the mixed tiling for B images of H x W; false: this map / batch keeps the uniform tiling
bool wino_mix_layout(int B, int H, int W, int Cin, int Cout, WinoPlanes* out);
// one kernel for the three transform roles of a mixed-tiling convolution `a` (whole images x 64 channels per workgroup):
//   M != null: Y = A^T M A of convolution `a` (+bias, activation, residual; out1 / out2 written when non-null), else the image comes from
//              a.in (per-channel affine in_scale / in_shift on in-image pixels when non-null);
//   Vnext != null: V = B^T d B for the next convolution (= `a` itself when M is null) from that image (feed_aff: through a.s2 / a.t2).
void launch_wino_mix(const ConvArgs& a, const WinoPlanes& pl, const float* M, float* Vnext, int feed_aff, const float* in_scale,
                     const float* in_shift, bool pack_next, hipStream_t s);
void wino_filter_transform(const double g[9], double u[36]);
void wino_debug_slots(int slots);            // test hook: workgroup slots the multi-tile Winograd GEMM sizes its grid for (0 = the device)
long wino_rows(long tiles);                  // rows per frequency plane of the V / M workspaces (tiles rounded up to 256)     // host: G g G^T of one 3x3 filter
int conv_wt_rows(int Cout);                   // packed weight rows (Cout rounded up to 128)
size_t conv_slab_floats();
// host: plan-layout weights [Cout][ks*ks][Cin] -> packed [conv_wt_rows(Cout)][conv_kpad(ks*ks*Cin)] (dst pre-zeroed)
void conv_pack_weights(const float* w, int Cout, int Cin, int ks, float* dst);                    // size of the stream-K slab workspace
inline int conv_kpad(int Ktot) { return (Ktot + 31) / 32 * 32; }

// --------------------------------------------------------------------------------------------
// Bandwidth-bound graph ops (ops_misc.hip)
// --------------------------------------------------------------------------------------------
// depthwise 3x3 pad 1 (+bias, ReLU / PReLU with per-channel slope); w9c = [9][C]
void launch_dwconv3x3(const float* in, const float* w9c, const float* bias, float* out, int B, int H, int W,
                      int C, int stride, int act, const float* slope, hipStream_t s);
// depthwise k x k VALID over a k x k map -> [B, C] (taps = k*k, w = [taps][C])
void launch_dwglobal(const float* in, const float* w, const float* bias, float* out, int B, int taps, int C, int act, const float* slope,
                     hipStream_t s);
// grouped 3x3 pad 1 with G = 2 | 4 channels per group on both sides; w = [9][C][G]
void launch_gconv3x3(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, int C, int G, int stride, int act,
                     const float* slope, hipStream_t s);
void launch_affine(const float* in, const float* sc, const float* sh, float* out, long pixels, int C, hipStream_t s);
void launch_act(const float* in, const float* slope, float* out, long pixels, int C, int act, hipStream_t s);
void launch_add(const float* a, const float* b, float* out, long n, hipStream_t s);
void launch_upsample2x(const float* in, float* out, int B, int H, int W, int C, hipStream_t s);

// --------------------------------------------------------------------------------------------
// Non-NN kernels of the face path (face_kernels.hip)
// --------------------------------------------------------------------------------------------
struct FaceRec {            // POD mirror of reference struct FaceBox (src/face_detector.h:8-12), 60 B
    int32_t x, y, w, h;
    float score;
    float lm[10];
};

// FaceDetector::preprocess (src/face_detector.cpp:92-137) → NHWC4 fp32 [B,inH,inW,4]
//   frames: B images, each rows x cols BGR u8 with row pitch `step` bytes and image pitch `img_stride`.
void launch_det_preprocess(const uint8_t* frames, long img_stride, int rows, int cols, int step, int B,
                           int inH, int inW, int newH, int newW, float* out, hipStream_t s);
// FaceRecognizer::preprocess (src/face_recognizer.cpp:135-150) on aligned crops → NHWC4
void launch_rec_preprocess(const uint8_t* crops, int n, int H, int W, float* out, hipStream_t s);
// preprocess fused into the graph's first Conv 3x3 (Cin = 3): u8 BGR image -> [B,Ho,Wo,Cout] fp32.
// w27 = [27][Cout] with k = (ky*3+kx)*3 + ci (ci in RGB order); act as fh::Act.
// wf / biasf (optional): the same filter with the normalisation folded in, for the thread-per-pixel kernel: wf[(tap*3 + j)][Cout] =
// w27[tap*3 + (2-j)] / 128 (j = byte of the BGR pixel), biasf = bias - 127.5/128 * sum_k w27[k].
// wfrag (optional, Cout % 16 == 0, Cout <= 64): the folded weights as bf16 MFMA fragments (stem_pack_wfrag) -> matrix-core stem kernel
void launch_stem_conv_u8(const uint8_t* src, long img_stride, int srcH, int srcW, int step, int B, int inH, int inW, int stride,
                         int Cout, const float* w27, const float* bias, const float* wf, const float* biasf, const float* slope, int act,
                         float* out1, float* out2, const float* s2, const float* t2, hipStream_t s, const unsigned* wfrag = nullptr);

struct DecodeArgs {
    const float* score[3];  // per stride [B, gh*gw*2]
    const float* bbox[3];   // [B, gh*gw*2, 4]
    const float* kps[3];    // [B, gh*gw*2, 10]
    int inH, inW, B;
    float scale, thr;
    FaceRec* cand;          // [B][cap]
    unsigned long long* keys;   // [B][cap]  sort keys (score desc, anchor index asc)
    int* count;             // [B]
    int cap;
};
void launch_scrfd_decode(const DecodeArgs& a, hipStream_t s);
// generic [rows, feat>=15] pre-decoded layout (src/face_detector.cpp:242-325)
void launch_rows_threshold(const float* rows, int B, int n, int feat, float scale, float thr, FaceRec* cand,
                           unsigned long long* keys, int* count, int cap, hipStream_t s);
// sort by key + greedy integer-IoU NMS (src/face_detector.cpp:340-384); writes survivors
// (score-descending) to out[B][max_out], counts to out_count[B].
void launch_sort_nms(const FaceRec* cand, unsigned long long* keys, const int* count, int cap, int B, float nms_thr,
                     FaceRec* out, int* out_count, int max_out, int* order_ws, hipStream_t s);

// FaceRecognizer::alignFace (src/face_recognizer.cpp:93-133): similarity estimate + warpAffine
//   faces[n] with frame index frame_of[n]; writes crops [n,112,112,3] BGR u8 and ok[n].
void launch_align(const uint8_t* frames, long img_stride, int rows, int cols, int step, const FaceRec* faces,
                  const int* frame_of, int n, int outH, int outW, uint8_t* crops, int* ok, hipStream_t s);
void launch_resize_u8c3(const uint8_t* src, long src_stride, int sh, int sw, int sstep, uint8_t* dst, long dst_stride,
                        int dh, int dw, int dstep, int n, hipStream_t s);

// FaceRecognizer::normalize (src/face_recognizer.cpp:306-318), one wave per row
void launch_l2_normalize(const float* in, float* out, int n, int dim, hipStream_t s);

// compareFaces generalised to 1:N (src/face_recognizer.cpp:320-334): top-k of (dot+1)/2 ranked (score desc, gallery index asc).
// gallery.hip: ONE streaming pass — gallery rows x queries on the matrix cores, per-workgroup top-k lists [gallery_parts][Q][k] kept in
// LDS while the rows go by (no G x Q matrix in memory); then launch_topk_merge.  qpacked = [ceil64(Q)][dim], zero rows behind Q.
int gallery_parts(long G, int Q, int* tiles_per_part);
void launch_gallery_topk(const float* gal, long G, int dim, const float* qpacked, int Q, int k, long idx_base, float* part_score, int* part_idx,
                         float* seed_score, int* seed_idx, hipStream_t s);      // seed_*: [Q][k] scratch (threshold pre-pass of large galleries)
void launch_label(const float* best_score, const int* best_idx, int n, float thr, int* labels, hipStream_t s);
void launch_topk_merge(const float* part_score, const int* part_idx, int nparts, int Q, int k, float* out_score,
                       int* out_idx, hipStream_t s);
void launch_topk_merge_strided(const float* part_score, const int* part_idx, int nparts, int Q, int k, long part_stride, float* out_score,
                               int* out_idx, hipStream_t s);

// detect -> embed hand-off: first min(count,F) faces per frame, densely packed (n <= 4096 frames)
void launch_select_faces(const FaceRec* det, const int* counts, int n, int per_frame, int F, FaceRec* faces, int* frame_of,
                         int* total, hipStream_t s);

}  // namespace fh
