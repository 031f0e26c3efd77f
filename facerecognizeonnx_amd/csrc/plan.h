// plan.h — ONNX graph → fused channels-last (NHWC) execution plan for the closed op set of the
// two face graphs (SCRFD det_500m, ArcFace IResNet): Conv (dense / depthwise), BatchNormalization,
// PRelu, Relu, Sigmoid, Add, Resize(nearest x2), Transpose(0,2,3,1), Reshape, Flatten, Gemm.
//
// This is host-only code (no HIP): it replaces ONNX Runtime's graph optimiser
// (ORT_ENABLE_ALL, reference src/face_detector.cpp:11) for this path.
#pragma once
#include <string>
#include <vector>

#include "onnx_reader.h"

namespace fh {

enum class Act : int { NONE = 0, RELU = 1, PRELU = 2, SIGMOID = 3 };
enum class ResMode : int { NONE = 0, SAME = 1, UP2X = 2 };
// DWGLOBAL: depthwise k x k VALID convolution over a k x k map (MobileFaceNet's GDC) -> 1 x 1; weight [k*k][C].
// GCONV: grouped 3x3 with 2 or 4 channels per group on both sides (MobileFaceNet's second layer); weight [9][Cout][G].
enum class OpKind : int { CONV = 0, DWCONV = 1, GEMM = 2, AFFINE = 3, ACT = 4, ADD = 5, UPSAMPLE = 6, DWPW = 7, DWGLOBAL = 8, GCONV = 9 };

struct PTensor {
    std::string name;
    int H = 1, W = 1, C = 1;          // stored NHWC extent per image (graph input C=3 is stored as 4)
    size_t elems() const { return (size_t)H * W * C; }
    size_t offset = 0;                // per-image float offset inside the activation arena
    int first = -1, last = -1;        // op index range in which the tensor is live
    bool is_input = false, is_output = false;
};

struct POp {
    OpKind kind = OpKind::CONV;
    std::string name;
    int in = -1, in2 = -1;            // in2: second operand of a standalone ADD
    int out = -1, out2 = -1, res = -1;
    ResMode res_mode = ResMode::NONE;
    Act act = Act::NONE;
    int ks = 1, stride = 1, pad = 0;
    int Cin = 0, Cout = 0, H = 0, W = 0, Ho = 0, Wo = 0;
    // Parameters (host fp32):
    //  CONV  : weight[Cout][ks*ks][Cin]   (k = tap*Cin + ci, Cin = stored channels of `in`)
    //  DWCONV: weight[9][C]
    //  GEMM  : weight[N][K], K re-ordered to the NHWC flatten of the producer
    //  AFFINE: s2/t2 applied to `in`
    std::vector<float> weight, bias, slope, s2, t2;
    // Horizontally merged sibling convolutions (same input, same geometry, sum(Cout) <= 32, e.g. the
    // SCRFD cls/reg/kps branches): one GEMM with Cout = sum, channel range [out_c0[g], out_c0[g+1]) goes
    // to tensor outs[g] with activation out_act[g].  Empty = ordinary single-output conv.
    std::vector<int> outs, out_c0, out_act;
    // DWPW: a depthwise 3x3 of stride dw_stride (dw_weight [9][Cin], dw_bias, dw_act) feeding this op's 1x1
    // convolution; H x W = the depthwise INPUT, Ho x Wo = its output = the pointwise grid
    std::vector<float> dw_weight, dw_bias;
    Act dw_act = Act::NONE;
    int dw_stride = 1;
    int weight_group = 1;            // GCONV: channels per group (2 | 4)
    // 3x3 stride-1 pad-1 CONV whose input is the BatchNorm'ed second output of op[bn_src] and which is that tensor's only reader:
    // the engine's Winograd input transform may read op[bn_src].out instead and apply s2 / t2 itself (exact: padding stays
    // zero), so that the second output is never written.  The planner keeps op[bn_src].out alive up to this op for that.
    int bn_src = -1;
    // 3x3 CONV (no activation) whose residual is the output of op[sc_src], a 1x1 convolution with no other reader (an IResNet block's
    // strided shortcut): the engine may run the shortcut as a tenth tap of this op's K loop (weights concatenated along K, biases
    // summed) and skip op[sc_src].  The planner keeps op[sc_src]'s INPUT alive up to this op for that.
    int sc_src = -1;
    double macs = 0;                  // multiply-accumulates per image
    double bytes = 0;                 // algorithmic activation bytes per image (in + res + outs)
};

struct OutDesc {
    std::string name;
    int tensor = -1;
    int rows = 0, cols = 0;           // per-image view: rows x cols floats, contiguous
};

struct Plan {
    std::vector<PTensor> tensors;
    std::vector<POp> ops;
    int input = -1;
    int inH = 0, inW = 0, inC = 3;
    std::vector<OutDesc> outputs;     // in graph output order
    size_t arena_elems = 0;           // per image
    double macs = 0, act_bytes = 0, weight_bytes = 0;
    std::string describe() const;     // human-readable op list (tests + DESIGN.md tables)
};

// inH/inW: the spatial size to plan for — the model's static shape if it has one, else the
// reference's defaults 640 / 112 (src/face_detector.cpp:39-57).
Plan build_plan(const OnnxModel& m, int inH, int inW);

}  // namespace fh
