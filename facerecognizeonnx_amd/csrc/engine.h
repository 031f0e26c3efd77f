// engine.h — device-side owner of one loaded graph (weights + activation arena + launch
// sequence) and the detector / recognizer / gallery objects built on it.  These are the
// objects behind the opaque C handles of include/facehip.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"
#include "plan.h"

namespace fh {


struct DevBuf {                // owning hipMalloc buffer
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf();
    void ensure(size_t n);     // grow-only (re)allocation
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

unsigned long long layout_epoch();   // bumped by every DevBuf (re)allocation / release in the process (key of captured graphs)

// One ONNX graph planned for a fixed input size, resident on the current device.
class Net {
  public:
    Net(const std::string& onnx_path, int default_h, int default_w);
    void reserve(int max_batch);                          // arena + split-K slabs for this batch
    float* input() const { return arena_.as<float>() + plan_.tensors[plan_.input].offset * (size_t)cap_; }
    float* output(int i) const { return tensor_ptr(plan_.outputs[i].tensor); }
    void run(int batch, hipStream_t s, int first_op = 0);
    // preprocess (+ first conv when it can be fused) straight from BGR u8 images, then the rest of the graph.
    // srcH x srcW = pasted image (<= net input; the remainder is the zero letterbox canvas)
    void run_u8(const uint8_t* src, long img_stride, int srcH, int srcW, int step, int batch, hipStream_t s);
    bool fuse_stem = true;                                // tuning / test hook: keep the preprocessed input tensor
    bool fuse_front = true;                               // tuning / test hook: stem conv inside the first depthwise -> pointwise kernel
    const Plan& plan() const { return plan_; }
    int in_h() const { return plan_.inH; }
    int in_w() const { return plan_.inW; }
    int capacity() const { return cap_; }
    bool winograd = true;                                 // Winograd F(4x4,3x3) for the deep 3x3 convs (false: direct form everywhere)
    bool halo_conv = true;                                // tuning / test hook: spatial-tile kernel for the thin 3x3 convolutions
    bool fold_shortcut = true;                            // tuning / test hook: a block's strided 1x1 shortcut as a tenth tap of the 3x3 it is added to
    bool fuse_wino = true;                                // tuning / test hook: fused output+input transform between consecutive Winograd layers
    // opt-in precision mode (fh_rec_set_precision): the Winograd GEMMs take split-bf16 operands (hi + mid bf16 per value, three bf16 MFMAs,
    // f32 accumulate); transforms, epilogues and every other layer stay fp32.  Returns the number of layers that switch.
    int set_bf16x2(bool on, hipStream_t s);
    bool bf16x2() const { return bf16x2_; }
    int cus = 0;                                          // CUs of the stream this net runs on when it is CU-masked (0 = all)
    int force_cfg = -1;                                   // tuning hook: conv tile config override
    bool sk_enable = true;                                // tuning hook: stream-K remainder wave
    // every switch above that changes which kernels run or with which arguments, as one word: part of the key of a captured
    // batch-1 graph (api.cpp), so a setter between two calls can never be answered by a replay of the old launch sequence
    long long config_word() const {
        return (long long)fuse_stem | (long long)fuse_front << 1 | (long long)winograd << 2 | (long long)halo_conv << 3 |
               (long long)fold_shortcut << 4 | (long long)fuse_wino << 5 | (long long)bf16x2_ << 6 | (long long)sk_enable << 7 |
               (long long)(cus & 0xffff) << 8 | (long long)((force_cfg + 1) & 0xff) << 24;
    }

  private:
    struct DevOp {
        size_t wt = 0, bias = 0, slope = 0, s2 = 0, t2 = 0;   // float offsets into params_
        bool has_slope = false, has_aff = false;
        size_t w27 = 0;                                       // stem layout [27][Cout] (op 0 only)
        size_t wfr = 0;                                       // ... and wf as bf16 MFMA fragments (stem_pack_wfrag), 16-channel stems
        size_t wf = 0, bf = 0;                                // ... and with the u8 normalisation folded in (byte order, w / 128; adjusted bias)
        size_t dww = 0, dwb = 0;                              // fused depthwise front end (DWPW)
        size_t w36 = 0;                                       // Winograd F(4,3) weights U[36][rows][Cin]
        size_t w36n = 0, w36p = 0;                            // ... their float count; offset of the split-bf16 copy in w36_bf_ (bf2 layers)
        bool bf2 = false;                                     // this layer's GEMM has a split-bf16 form
        size_t wfrag = 0;                                     // halo-conv weights in MFMA fragment order (conv_halo.hip)
        bool halo = false;                                    // eligible for the spatial-tile 3x3 kernel
        bool wino = false;                                    // eligible: 3x3 stride 1 pad 1, Cin >= 128
        size_t w2 = 0;                                        // fused Winograd F(2x2,3x3) image of the filter (conv_wino2.hip): 3x3 stride 1 pad 1,
        bool w2ok = false;                                    //   32 <= Cin < 128, Cout % 64 == 0 (IResNet's 64-channel stages)
        int aff_src = -1;                                     // Winograd op whose input is op[aff_src]'s second (BatchNorm) output and its only
                                                              // consumer: the transform reads op[aff_src].out and applies that affine itself
        int aff_dst = -1;                                     // ... and the producer's side of the same link
        int bn_fold_src = -1;                                 // direct 3x3 conv with its block's BatchNorm folded in (weights * s, 9-class bias):
                                                              // reads op[bn_fold_src].out, whose second output is then never written
        bool bn_fold_dst = false;                             // ... and the producer's side: skip out2
        int sc_src = -1;                                      // 3x3 conv that can take op[sc_src] (1x1 shortcut) into its K loop ...
        size_t wt_sc = 0, bias_sc = 0;                        // ... weights [rows][9*Cin + sc_C] and summed bias for that form
        int Kpad_sc = 0;
        bool sc_dst = false;                                  // ... and the shortcut's side: skipped when its consumer folds it
        bool fuse_next = false;                               // Winograd op followed by another on the same small map: fused transform kernel
        bool fuse_feed_aff = false;                           //   the next conv sees out * s2 + t2 (its block's BatchNorm) instead of out
        bool fuse_keep_out1 = true;                           //   something else reads the plain output too (e.g. a later residual): write it
        int Kpad = 0;
    };
  public:
    float* workspace() const { return partial_.as<float>(); }   // stream-K slabs (diagnostic builds of dwpw_mfma.hip park phase stamps there)
    unsigned* error_record() const { return sk_rec_.p; }        // this Net's stream-K watchdog record (null before reserve)
  private:
    float* tensor_ptr(int t) const { return arena_.as<float>() + plan_.tensors[t].offset * (size_t)cap_; }
    Plan plan_;
    std::vector<DevOp> dev_;
    DevBuf params_, arena_, partial_, wino_v_, wino_m_, w36_bf_;
    bool bf16x2_ = false;
    size_t wino_maxc_ = 0;
    size_t wino_elems_ = 0;                                   // per image: 36 * tiles * max(Cin, Cout) of the largest Winograd op
    int cap_ = 0;
    struct SkRecord {                                         // host-mapped; parked on a free list at destruction (the device may still write it)
        unsigned* p = nullptr;
        int device = -1;                                      // the device whose launches write it (set with p)
        SkRecord() = default;
        SkRecord(const SkRecord&) = delete;
        SkRecord& operator=(const SkRecord&) = delete;
        ~SkRecord();
    } sk_rec_;
    unsigned sk_gen_ = 0;                                     // conv_error_generation() this Net's hand-off counters were last zeroed under
    bool stem_ok_ = false;
    bool front_ok_ = false;                                   // ops 0 + 1 = stem conv (16 channels) -> DW+PW: one kernel
};

class Detector {
  public:
    explicit Detector(const std::string& onnx_path);
    // frames: device pointer, n images of rows x cols BGR u8 (row pitch `step`, image pitch `stride`)
    // out: device [n][max_out] FaceRec, counts: device [n].  Asynchronous on stream s.
    void detect_dev(const uint8_t* frames, int n, int rows, int cols, int step, long stride, float score_thr, float nms_thr,
                    FaceRec* out, int max_out, int* counts, hipStream_t s);
    // network + decode only, for tests: rows15 = device [n][N][15]? not needed — outputs are read through net()
    Net& net() { return net_; }
    int num_anchors() const { return anchors_; }
    bool predecoded() const { return predecoded_; }
    float last_scale() const { return scale_; }
    // stage hooks used by parity tests (device pointers)
    void run_network_dev(const uint8_t* frames, int n, int rows, int cols, int step, long stride, hipStream_t s);
    void postprocess_dev(int n, float score_thr, float nms_thr, FaceRec* out, int max_out, int* counts, hipStream_t s);

  private:
    void reserve(int n, int rows, int cols);
    Net net_;
    bool predecoded_ = false;
    int anchors_ = 0, feat_ = 15, cap_ = 0, nb_ = 0;
    float scale_ = 1.f;
    DevBuf resized_, cand_, keys_, count_, ws_;
};

class Recognizer {
  public:
    explicit Recognizer(const std::string& onnx_path);
    int dim() const { return dim_; }
    Net& net() { return net_; }
    // aligned crops [n][H][W][3] BGR u8 (device) -> L2-normalised embeddings [n][dim] (device)
    void embed_aligned_dev(const uint8_t* crops, int n, float* out, hipStream_t s, float* raw_out = nullptr);
    // alignFace + embed: faces[n] (device) on frames; ok[n] (device, may be null) 1/2 = produced, 0 = empty
    void embed_faces_dev(const uint8_t* frames, int rows, int cols, int step, long stride, const FaceRec* faces,
                         const int* frame_of, int n, float* out, int* ok, hipStream_t s);
    void align_dev(const uint8_t* frames, int rows, int cols, int step, long stride, const FaceRec* faces, const int* frame_of,
                   int n, uint8_t* crops, int* ok, hipStream_t s);
    void resize_embed_dev(const uint8_t* frames, int n, int rows, int cols, int step, long stride, float* out, hipStream_t s);
    int max_chunk = 256;                                 // faces per network pass

  private:
    Net net_;
    int dim_ = 0;
    DevBuf crops_, ok_, raw_;
};

class Gallery {
  public:
    explicit Gallery(int dim) : dim_(dim) {}
    void upload(const float* rows, long n, bool device_src, long index_base);
    long enroll(const float* rows, long n, bool device_src);     // append; returns the global index of the first new row
    // best row per query if its mapped score > thr, else -1 (main.cpp:229-233); out_score = that best score
    void label_dev(const float* q, int Q, float thr, int* out_label, float* out_score, hipStream_t s);
    // queries [Q][dim] device, Q <= 256, k <= 16 -> out_score/out_idx [Q][k] device
    void topk_dev(const float* q, int Q, int k, float* out_score, int* out_idx, hipStream_t s);
    long size() const { return n_; }
    int dim() const { return dim_; }

  private:
    int dim_;
    long n_ = 0, base_ = 0;
    DevBuf rows_, qpack_, ps_, pi_, best_i_, seed_s_, seed_i_;
};

}  // namespace fh
