// ops_misc.hip — the bandwidth-bound graph operators of the SCRFD / IResNet plans that are not
// MFMA work: depthwise 3x3 convolution (+bias +ReLU), stand-alone BatchNormalization (affine),
// stand-alone activations, Add and nearest 2x up-sampling.  All tensors are channels-last fp32,
// every lane moves 16 bytes (4 channels) per access so a wave covers whole 128-byte lines.
// These are the ONNX nodes ORT runs inside session_->Run (reference src/face_detector.cpp:179-183).
// Also here: the first convolution fused with the u8 preprocess (stem_conv_u8_kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "kernels.h"
#include "plan.h"
#include "stem_fma.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));

static inline int grid_for(long n, int block = 256, int cap = 256 * 16) {
    long b = (n + block - 1) / block;
    return (int)(b < 1 ? 1 : b > cap ? cap : b);
}

__device__ __forceinline__ float act1(float v, int act, float slope) {
    if (act == (int)Act::RELU) return v > 0.f ? v : 0.f;
    if (act == (int)Act::PRELU) return v >= 0.f ? v : v * slope;
    if (act == (int)Act::SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// Depthwise 3x3 (+bias +ReLU).  One thread = 4 channels x a VERTICAL strip of STRIP output pixels:
// the (STRIP*stride + 2) x 3 input float4s are loaded once and reused down the strip (4.5 instead
// of 9 loads per output at stride 1), weights stay in registers.  Lanes run over the channel groups
// and then over x, so every load / store instruction of a wave touches one contiguous run of pixels
// (full 128-byte lines); a horizontal strip would scatter each instruction over 16 half-used lines.
template <int STRIDE, int STRIP>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int B, int H, int W, int C, int Ho, int Wo, int act,
                                                        const float* __restrict__ slope) {
    constexpr int ROWS = (STRIP - 1) * STRIDE + 3;
    const int C4 = C >> 2;
    const int strips = (Ho + STRIP - 1) / STRIP;
    const long total = (long)B * strips * Wo * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        long r = idx / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int sy = (int)(r % strips);
        const int n = (int)(r / strips);
        const int oy0 = sy * STRIP;
        v4f wk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const v4f*>(w + t * C + c4 * 4);
        const v4f b4 = *reinterpret_cast<const v4f*>(bias + c4 * 4);
        const v4f sl4 = slope ? *reinterpret_cast<const v4f*>(slope + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
        v4f acc[STRIP];
#pragma unroll
        for (int o = 0; o < STRIP; ++o) acc[o] = b4;
        const int iy0 = oy0 * STRIDE - 1, ix0 = ox * STRIDE - 1;
        // UNCONDITIONAL loads from clamped coordinates, zeroed by a select afterwards: a load under a branch is waited for at the
        // branch's join — three latency chains of ROWS loads per thread instead of 3 x ROWS loads in flight (same rule as wino_input_kernel)
        const float* imgp = in + (size_t)n * H * W * C + c4 * 4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ix0 + kx;
            const bool okx = (unsigned)ix < (unsigned)W;
            const int ixc = min(max(ix, 0), W - 1);
            v4f x[ROWS];
#pragma unroll
            for (int ridx = 0; ridx < ROWS; ++ridx) {
                const int iy = iy0 + ridx;
                const bool ok = okx && (unsigned)iy < (unsigned)H;
                const v4f ld = *reinterpret_cast<const v4f*>(imgp + ((size_t)min(max(iy, 0), H - 1) * W + ixc) * C);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[ridx][e] = ok ? ld[e] : 0.f;   // a select (v_cndmask), not a multiply: the clamped neighbour may be Inf / NaN
            }
#pragma unroll
            for (int o = 0; o < STRIP; ++o)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) acc[o] += x[o * STRIDE + ky] * wk[ky * 3 + kx];
        }
#pragma unroll
        for (int o = 0; o < STRIP; ++o) {
            const int oy = oy0 + o;
            if (oy >= Ho) break;
            v4f v = acc[o];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act1(v[e], act, sl4[e]);
            *reinterpret_cast<v4f*>(out + (((size_t)n * Ho + oy) * Wo + ox) * C + c4 * 4) = v;
        }
    }
}

// Horizontal-strip variant (strip along x): better for stride 2, where a vertical strip would make every
// wave instruction skip every other pixel.
template <int STRIDE, int STRIP>
__global__ __launch_bounds__(256) void dwconv3x3_hstrip_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int B, int H, int W, int C, int Ho, int Wo, int act,
                                                        const float* __restrict__ slope) {
    constexpr int COLS = (STRIP - 1) * STRIDE + 3;
    const int C4 = C >> 2;
    const int strips = (Wo + STRIP - 1) / STRIP;
    const long total = (long)B * Ho * strips * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        long r = idx / C4;
        const int sx = (int)(r % strips); r /= strips;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const int ox0 = sx * STRIP;
        v4f wk[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const v4f*>(w + t * C + c4 * 4);
        const v4f b4 = *reinterpret_cast<const v4f*>(bias + c4 * 4);
        const v4f sl4 = slope ? *reinterpret_cast<const v4f*>(slope + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
        v4f acc[STRIP];
#pragma unroll
        for (int o = 0; o < STRIP; ++o) acc[o] = b4;
        const int iy0 = oy * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
        const float* imgp = in + (size_t)n * H * W * C + c4 * 4;           // (unconditional clamped loads: see dwconv3x3_kernel)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = iy0 + ky;
            const bool oky = (unsigned)iy < (unsigned)H;
            const float* rowp = imgp + (size_t)min(max(iy, 0), H - 1) * W * C;
            v4f x[COLS];
#pragma unroll
            for (int cidx = 0; cidx < COLS; ++cidx) {
                const int ix = ix0 + cidx;
                const bool ok = oky && (unsigned)ix < (unsigned)W;
                const v4f ld = *reinterpret_cast<const v4f*>(rowp + (size_t)min(max(ix, 0), W - 1) * C);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[cidx][e] = ok ? ld[e] : 0.f;
            }
#pragma unroll
            for (int o = 0; o < STRIP; ++o)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[o] += x[o * STRIDE + kx] * wk[ky * 3 + kx];
        }
#pragma unroll
        for (int o = 0; o < STRIP; ++o) {
            const int ox = ox0 + o;
            if (ox >= Wo) break;
            v4f v = acc[o];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act1(v[e], act, sl4[e]);
            *reinterpret_cast<v4f*>(out + (((size_t)n * Ho + oy) * Wo + ox) * C + c4 * 4) = v;
        }
    }
}

// Lean form (round 4).  The two kernels above spend their time on instructions, not on memory: ~600 per thread (64-bit index arithmetic
// with four divisions, four selects per loaded float4, an activation switch that carries the sigmoid's exp / divide) for 36 useful float4
// FMAs — 20x20x288 at B = 128 ran 31 us against a ~15-20 us traffic floor (in + out = 118 MB, both cache-resident).  Here:
//   * grid = (output row run, strip of rows, image): image and rows are scalars; a thread is float4 column r of the output row run
//     [Wo][C / 4]; its centre tap is input column r (stride 1) or 2 r - c4 (stride 2), the neighbours are -+ C/4 from there, so no pixel
//     coordinate is ever computed (the channel group c4 = r mod C/4 comes from one multiply-high);
//   * the taps come in through BUFFER loads with a per-image descriptor: rows above / below the image are out of range and read as zero
//     in hardware; the left / right column is pushed out of range by ONE select on its offset (not four per load);
//   * activation is a template parameter.
// Same summation order as the generic kernels (stride 1: columns left to right, rows inside; stride 2: rows top to bottom, columns
// inside): bit-identical results.
constexpr unsigned DW_OOB = 0x40000000u;                                     // an offset beyond any image (images are < 1 GB: host check)
template <int STRIDE, int STRIP, int ACT>
__global__ __launch_bounds__(256) void dwconv3x3_lean_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out, int H, int Lin, int Ho,
                                                             int Lout, int C4, unsigned c4_magic, const float* __restrict__ slope) {
    constexpr int ROWS = (STRIP - 1) * STRIDE + 3;
    const int r = (int)(blockIdx.x * 256 + threadIdx.x);
    if (r >= Lout) return;
    const int n = (int)blockIdx.z, oy0 = (int)blockIdx.y * STRIP;
    const int c4 = r - (int)__umulhi((unsigned)r, c4_magic) * C4;           // r % C4 (c4_magic = ceil(2^32 / C4); exact for r < 2^20)
    const int C = C4 * 4;
    const size_t img = (size_t)H * Lin * 4;                                 // floats per input image
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * img), 0, (int)(img * 4), 0x00020000);
    const unsigned rowb = (unsigned)Lin * 16u;
    const int centre = STRIDE == 1 ? r : 2 * r - c4;
    unsigned vc[3];
    vc[1] = (unsigned)centre * 16u;
    vc[0] = r >= C4 ? vc[1] - (unsigned)C4 * 16u : DW_OOB;
    vc[2] = centre + C4 < Lin ? vc[1] + (unsigned)C4 * 16u : DW_OOB;
    const unsigned row0 = (unsigned)(oy0 * STRIDE - 1) * rowb;               // (oy0 = 0: wraps to "far out of range" together with any vc)
    v4f x[3][ROWS];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ri = 0; ri < ROWS; ++ri)
            x[kx][ri] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vc[kx] + row0 + (unsigned)ri * rowb, 0, 0));
    v4f wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *reinterpret_cast<const v4f*>(w + t * C + c4 * 4);
    const v4f b4 = *reinterpret_cast<const v4f*>(bias + c4 * 4);
    v4f acc[STRIP];
#pragma unroll
    for (int o = 0; o < STRIP; ++o) acc[o] = b4;
    if constexpr (STRIDE == 1) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int o = 0; o < STRIP; ++o)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) acc[o] += x[kx][o + ky] * wk[ky * 3 + kx];
    } else {
#pragma unroll
        for (int o = 0; o < STRIP; ++o)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[o] += x[kx][o * STRIDE + ky] * wk[ky * 3 + kx];
    }
    v4f sl4 = v4f{0.f, 0.f, 0.f, 0.f};
    if constexpr (ACT == (int)Act::PRELU) sl4 = *reinterpret_cast<const v4f*>(slope + c4 * 4);
    float* __restrict__ op = out + ((size_t)n * Ho + oy0) * Lout * 4 + (size_t)r * 4;
#pragma unroll
    for (int o = 0; o < STRIP; ++o) {
        if (oy0 + o >= Ho) break;
        v4f v = acc[o];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (ACT == (int)Act::RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
            else if constexpr (ACT == (int)Act::PRELU) v[e] = v[e] >= 0.f ? v[e] : v[e] * sl4[e];
        }
        *reinterpret_cast<v4f*>(op + (size_t)o * Lout * 4) = v;
    }
}

template <int STRIDE, int ACT>
static void launch_dwconv3x3_lean(const float* in, const float* w9c, const float* bias, float* out, int B, int H, int W, int C, int Ho, int Wo,
                                  const float* slope, hipStream_t s) {
    constexpr int STRIP = STRIDE == 1 ? 4 : 2;
    const int C4 = C / 4, Lin = W * C4, Lout = Wo * C4;
    const unsigned magic = (unsigned)((0x100000000ull + (unsigned)C4 - 1) / (unsigned)C4);
    hipLaunchKernelGGL((dwconv3x3_lean_kernel<STRIDE, STRIP, ACT>), dim3((unsigned)((Lout + 255) / 256), (unsigned)((Ho + STRIP - 1) / STRIP), (unsigned)B),
                       dim3(256), 0, s, in, w9c, bias, out, H, Lin, Ho, Lout, C4, magic, slope);
}
template <int STRIDE>
static void launch_dwconv3x3_lean_act(const float* in, const float* w9c, const float* bias, float* out, int B, int H, int W, int C, int Ho, int Wo,
                                      int act, const float* slope, hipStream_t s) {
    if (act == (int)Act::RELU) launch_dwconv3x3_lean<STRIDE, (int)Act::RELU>(in, w9c, bias, out, B, H, W, C, Ho, Wo, slope, s);
    else if (act == (int)Act::PRELU) launch_dwconv3x3_lean<STRIDE, (int)Act::PRELU>(in, w9c, bias, out, B, H, W, C, Ho, Wo, slope, s);
    else launch_dwconv3x3_lean<STRIDE, (int)Act::NONE>(in, w9c, bias, out, B, H, W, C, Ho, Wo, slope, s);
}

void launch_dwconv3x3(const float* in, const float* w9c, const float* bias, float* out, int B, int H, int W, int C,
                      int stride, int act, const float* slope, hipStream_t s) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    constexpr int STRIP = 4;
    // (row runs < 2^20 float4s: the multiply-high remainder; image < 1 GB: DW_OOB; grid y / z limits)
    static const bool lean_on = [] { const char* e = getenv("FACEHIP_DW_LEAN"); return !e || atoi(e) != 0; }();
    const bool lean = lean_on && (stride == 1 || stride == 2) && C >= 8 && (long)W * (C / 4) < (1L << 20) &&
                      (long)H * W * C * 4 + (long)W * C * 4 * 8 < (long)DW_OOB && B <= 65535 && Ho <= 65535 * 2 &&
                      (act == (int)Act::NONE || act == (int)Act::RELU || (act == (int)Act::PRELU && slope));
    if (lean) {
        if (stride == 1) launch_dwconv3x3_lean_act<1>(in, w9c, bias, out, B, H, W, C, Ho, Wo, act, slope, s);
        else launch_dwconv3x3_lean_act<2>(in, w9c, bias, out, B, H, W, C, Ho, Wo, act, slope, s);
        return;
    }
    if (stride == 1) {
        const long total = (long)B * ((Ho + STRIP - 1) / STRIP) * Wo * (C / 4);
        hipLaunchKernelGGL((dwconv3x3_kernel<1, STRIP>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, in, w9c, bias, out, B, H, W, C, Ho, Wo, act, slope);
    } else {
        const long total = (long)B * Ho * ((Wo + STRIP - 1) / STRIP) * (C / 4);
        hipLaunchKernelGGL((dwconv3x3_hstrip_kernel<2, STRIP>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, in, w9c, bias, out, B, H, W, C, Ho, Wo, act, slope);
    }
}

// Depthwise k x k VALID convolution over a k x k map -> one value per channel (MobileFaceNet's "GDC" layer):
// out[b][c] = bias[c] + sum_p w[p][c] * in[b][p][c].  Thread = 4 channels of one image; lanes walk the channels.
__global__ __launch_bounds__(256) void dwglobal_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ out, int B, int taps, int C, int act, const float* __restrict__ slope) {
    const int C4 = C >> 2;
    const long total = (long)B * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const long b = idx / C4;
        v4f acc = *reinterpret_cast<const v4f*>(bias + c4 * 4);
        const float* x = in + (size_t)b * taps * C + c4 * 4;
        for (int p = 0; p < taps; ++p) acc += *reinterpret_cast<const v4f*>(x + (size_t)p * C) * *reinterpret_cast<const v4f*>(w + (size_t)p * C + c4 * 4);
        const v4f sl4 = slope ? *reinterpret_cast<const v4f*>(slope + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = act1(acc[e], act, sl4[e]);
        *reinterpret_cast<v4f*>(out + (size_t)b * C + c4 * 4) = acc;
    }
}
void launch_dwglobal(const float* in, const float* w, const float* bias, float* out, int B, int taps, int C, int act, const float* slope,
                     hipStream_t s) {
    const long total = (long)B * (C / 4);
    if (total > 0) hipLaunchKernelGGL(dwglobal_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, w, bias, out, B, taps, C, act, slope);
}

// Grouped 3x3 convolution (pad 1, stride 1 | 2) with G = 2 or 4 channels per group on both sides (MobileFaceNet's second layer:
// 64 groups of 2).  A float4 of 4 consecutive output channels depends on exactly the same 4 input channels, so the kernel has the
// shape of the depthwise one with a G-wide dot product per tap.  w: [9][C][G].  Thread = 4 channels of one output pixel.
template <int G>
__global__ __launch_bounds__(256) void gconv3x3_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo, int stride, int act,
                                                       const float* __restrict__ slope) {
    const int C4 = C >> 2;
    const long total = (long)B * Ho * Wo * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        long r = idx / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int n = (int)(r / Ho);
        v4f acc = *reinterpret_cast<const v4f*>(bias + c4 * 4);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const v4f x = *reinterpret_cast<const v4f*>(in + (((size_t)n * H + iy) * W + ix) * C + c4 * 4);
                const float* wt = w + ((size_t)(ky * 3 + kx) * C + c4 * 4) * G;      // [4 output channels][G]
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < G; ++j) acc[e] += wt[e * G + j] * x[(e / G) * G + j];
            }
        }
        const v4f sl4 = slope ? *reinterpret_cast<const v4f*>(slope + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = act1(acc[e], act, sl4[e]);
        *reinterpret_cast<v4f*>(out + (((size_t)n * Ho + oy) * Wo + ox) * C + c4 * 4) = acc;
    }
}
void launch_gconv3x3(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, int C, int G, int stride, int act,
                     const float* slope, hipStream_t s) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long total = (long)B * Ho * Wo * (C / 4);
    if (total <= 0) return;
    if (G == 2) hipLaunchKernelGGL(gconv3x3_kernel<2>, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, in, w, bias, out, B, H, W, C, Ho, Wo, stride, act, slope);
    else hipLaunchKernelGGL(gconv3x3_kernel<4>, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, in, w, bias, out, B, H, W, C, Ho, Wo, stride, act, slope);
}

// generic per-channel pass, scalar channel indexing (C need not be a multiple of 4)
__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ in, const float* __restrict__ sc,
                                                     const float* __restrict__ sh, float* __restrict__ out, long total, int C) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        out[i] = in[i] * sc[c] + sh[c];
    }
}
void launch_affine(const float* in, const float* sc, const float* sh, float* out, long pixels, int C, hipStream_t s) {
    const long total = pixels * C;
    hipLaunchKernelGGL(affine_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, sc, sh, out, total, C);
}

__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ in, const float* __restrict__ slope,
                                                  float* __restrict__ out, long total, int C, int act) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float sl = slope ? slope[i % C] : 0.f;
        out[i] = act1(in[i], act, sl);
    }
}
void launch_act(const float* in, const float* slope, float* out, long pixels, int C, int act, hipStream_t s) {
    const long total = pixels * C;
    hipLaunchKernelGGL(act_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, slope, out, total, C, act);
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
void launch_add(const float* a, const float* b, float* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, out, n);
}

__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                         int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long pix = i / C;
        const int ox = (int)(pix % Wo); pix /= Wo;
        const int oy = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        out[i] = in[(((size_t)n * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
    }
}
void launch_upsample2x(const float* in, float* out, int B, int H, int W, int C, hipStream_t s) {
    const long total = (long)B * 4 * H * W * C;
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, B, H, W, C);
}


// ------------------------------------------------------------------------------------------
// First convolution fused with the image preprocess (FaceDetector::preprocess
// src/face_detector.cpp:120-136 / FaceRecognizer::preprocess src/face_recognizer.cpp:135-150 +
// the graph's first Conv 3x3, Cin = 3): reads the BGR u8 image once (1 byte per sample instead
// of a 4-byte float written and read back), normalises into an LDS tile and runs the 27-tap
// stencil on the vector ALU.  K = 27 is far too thin for the matrix cores and the layer is bound
// by its output write anyway.  Letterbox area (inside the net input, outside the pasted image)
// = u8 0 = -0.99609375 after normalisation; outside the net input = the conv's zero padding.
// Thread = 4 output channels x (TILE*TILE*4/Cout... ) pixels; weights live in registers.
// ------------------------------------------------------------------------------------------
constexpr int STEM_TILE = 16;

template <int STRIDE>
__global__ __launch_bounds__(256) void stem_conv_u8_kernel(const uint8_t* __restrict__ src, long img_stride, int srcH, int srcW, int step,
                                                           int inH, int inW, int Ho, int Wo, int Cout, const float* __restrict__ w27,
                                                           const float* __restrict__ bias, const float* __restrict__ slope, int act,
                                                           float* __restrict__ out1, float* __restrict__ out2,
                                                           const float* __restrict__ s2, const float* __restrict__ t2, int ntiles) {
    constexpr int IT = (STEM_TILE - 1) * STRIDE + 3;                 // input tile edge
    __shared__ float tile[IT * IT * 3];
    const int tid = threadIdx.x;
    const int tiles_x = (Wo + STEM_TILE - 1) / STEM_TILE, tiles_y = (Ho + STEM_TILE - 1) / STEM_TILE;
    const int G = Cout >> 2;                                         // channel groups of 4
    const int c4 = tid % G;
    const int pix_per_pass = 256 / G;
    v4f w[27];                                                       // [tap][ci] for this thread's 4 channels, loaded ONCE per block
#pragma unroll
    for (int k = 0; k < 27; ++k) w[k] = *reinterpret_cast<const v4f*>(w27 + (size_t)k * Cout + c4 * 4);
    const v4f b4 = *reinterpret_cast<const v4f*>(bias + c4 * 4);
    v4f sl = {0.f, 0.f, 0.f, 0.f}, sc2 = {0.f, 0.f, 0.f, 0.f}, sh2 = {0.f, 0.f, 0.f, 0.f};
    if (slope) sl = *reinterpret_cast<const v4f*>(slope + c4 * 4);
    if (out2) { sc2 = *reinterpret_cast<const v4f*>(s2 + c4 * 4); sh2 = *reinterpret_cast<const v4f*>(t2 + c4 * 4); }
    // persistent blocks: each walks over many tiles, so the 27 weight vectors are fetched once
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int b = tl / (tiles_x * tiles_y);
        const int t = tl - b * tiles_x * tiles_y;
        const int oy0 = (t / tiles_x) * STEM_TILE, ox0 = (t % tiles_x) * STEM_TILE;
        const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
        const uint8_t* img = src + (size_t)b * img_stride;
        __syncthreads();                                             // previous tile fully consumed
        for (int i = tid; i < IT * IT; i += 256) {
            const int ty = i / IT, tx = i - ty * IT;
            const int iy = iy0 + ty, ix = ix0 + tx;
            float r = 0.f, g = 0.f, bl = 0.f;                        // conv zero padding
            if ((unsigned)iy < (unsigned)inH && (unsigned)ix < (unsigned)inW) {
                float vb = 0.f, vg = 0.f, vr = 0.f;                  // letterbox canvas = u8 zeros
                if (iy < srcH && ix < srcW) {
                    const uint8_t* p = img + (size_t)iy * step + (size_t)ix * 3;
                    vb = (float)p[0]; vg = (float)p[1]; vr = (float)p[2];
                }
                r = (vr - 127.5f) / 128.0f; g = (vg - 127.5f) / 128.0f; bl = (vb - 127.5f) / 128.0f;
            }
            tile[i * 3 + 0] = r; tile[i * 3 + 1] = g; tile[i * 3 + 2] = bl;   // RGB order = graph channel order
        }
        __syncthreads();
        if (tid >= G * pix_per_pass) continue;
        for (int p = tid / G; p < STEM_TILE * STEM_TILE; p += pix_per_pass) {
            const int py = p / STEM_TILE, px = p - py * STEM_TILE;
            const int oy = oy0 + py, ox = ox0 + px;
            if (oy >= Ho || ox >= Wo) continue;
            v4f acc = b4;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float* tp = tile + ((py * STRIDE + ky) * IT + px * STRIDE + kx) * 3;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) acc += w[(ky * 3 + kx) * 3 + ci] * tp[ci];
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[e];
                if (act == 1) v = v > 0.f ? v : 0.f;
                else if (act == 2) v = v >= 0.f ? v : v * sl[e];
                else if (act == 3) v = 1.0f / (1.0f + expf(-v));
                acc[e] = v;
            }
            const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * Cout + c4 * 4;
            if (out1) *reinterpret_cast<v4f*>(out1 + o) = acc;
            if (out2) *reinterpret_cast<v4f*>(out2 + o) = acc * sc2 + sh2;
        }
    }
}

// ---- thread-per-pixel form ------------------------------------------------------------------------------------------------
// One thread = one output pixel x ALL output channels.  The 27 input bytes of the pixel's 3x3 window are fetched as 3 aligned dwords
// per image row straight from the u8 frame (neighbouring lanes overlap: L1 / L2 serve the re-reads, each byte leaves HBM once),
// converted with v_cvt_f32_ubyteN, and multiplied by weights that live in SGPRs (every lane uses the same 27 x COUT weights: uniform
// loads through the scalar cache, one SGPR operand per v_fmac).  No LDS, no barrier, ~60 VGPRs: 8 waves per SIMD hide the loads.
// The normalisation (v - 127.5) / 128 is folded into the weights on the host: wf = w / 128 (a power of two: exact) in BYTE order
// (B, G, R per pixel), biasf = bias - 127.5/128 * sum(w).  That only works where all nine taps are real image pixels, so this kernel
// covers the INTERIOR of the output map; the thin frame of pixels whose window touches the zero padding, the letterbox canvas or the
// first / last pixel of a row goes to stem_conv_border_kernel (literal arithmetic, per-tap bounds).
typedef const unsigned __attribute__((address_space(1))) gmem_u32;
template <int STRIDE, int COUT>
__global__ __launch_bounds__(256) void stem_conv_px_kernel(const uint8_t* __restrict__ src, long img_stride, int step, int Ho, int Wo, int x0,
                                                           int y0, int nx, int ny, int B, const float* __restrict__ wf,
                                                           const float* __restrict__ biasf, const float* __restrict__ slope, int act,
                                                           float* __restrict__ out1, float* __restrict__ out2, const float* __restrict__ s2,
                                                           const float* __restrict__ t2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * ny * nx) return;
    const int ox = x0 + idx % nx;
    const int r0 = idx / nx;
    const int oy = y0 + r0 % ny, b = r0 / ny;
    const uint8_t* p = src + (size_t)b * img_stride + (size_t)(oy * STRIDE - 1) * step + (size_t)(ox * STRIDE - 1) * 3;
    float v[27];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const unsigned long long a = (unsigned long long)(p + (size_t)r * step);
        const unsigned sh = (unsigned)a & 3u;
        const gmem_u32* q = (const gmem_u32*)(a - sh);                       // (explicit global address space: no flat loads)
        const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
        const unsigned n0 = __builtin_amdgcn_alignbyte(d1, d0, sh), n1 = __builtin_amdgcn_alignbyte(d2, d1, sh), n2 = d2 >> (8u * sh);
        v[r * 9 + 0] = (float)(n0 & 255u); v[r * 9 + 1] = (float)((n0 >> 8) & 255u); v[r * 9 + 2] = (float)((n0 >> 16) & 255u); v[r * 9 + 3] = (float)(n0 >> 24);
        v[r * 9 + 4] = (float)(n1 & 255u); v[r * 9 + 5] = (float)((n1 >> 8) & 255u); v[r * 9 + 6] = (float)((n1 >> 16) & 255u); v[r * 9 + 7] = (float)(n1 >> 24);
        v[r * 9 + 8] = (float)(n2 & 255u);
    }
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = biasf[c];
    // weights in SGPRs, packed FMAs: see stem_fma.h.  One asm block per image row (9 taps) and 16-channel group.
#pragma unroll
    for (int g = 0; g < COUT / 16; ++g) {
        fh_v2f a2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a2[i] = fh_v2f{acc[g * 16 + 2 * i], acc[g * 16 + 2 * i + 1]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* wrow = wf + (size_t)(r * 9) * COUT + g * 16;
            fh_v2f xp[5];
#pragma unroll
            for (int i = 0; i < 4; ++i) xp[i] = fh_v2f{v[r * 9 + 2 * i], v[r * 9 + 2 * i + 1]};
            xp[4] = fh_v2f{v[r * 9 + 8], 0.f};
#if defined(__HIP_DEVICE_COMPILE__)
            FH_STEM_ROW_FMA(a2, xp, wrow, COUT * 4);
#else
            (void)wrow; (void)xp;
#endif
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[g * 16 + 2 * i] = a2[i][0]; acc[g * 16 + 2 * i + 1] = a2[i][1]; }
    }
    const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * COUT;
#pragma unroll
    for (int c4 = 0; c4 < COUT / 4; ++c4) {
        v4f y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = acc[c4 * 4 + e];
            if (act == 1) u = u > 0.f ? u : 0.f;
            else if (act == 2) u = u >= 0.f ? u : u * slope[c4 * 4 + e];
            else if (act == 3) u = 1.0f / (1.0f + expf(-u));
            y[e] = u;
        }
        if (out1) *reinterpret_cast<v4f*>(out1 + o + c4 * 4) = y;
        if (out2) {
            v4f z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = y[e] * s2[c4 * 4 + e] + t2[c4 * 4 + e];
            *reinterpret_cast<v4f*>(out2 + o + c4 * 4) = z;
        }
    }
}

// Frame of the output map around the interior [x0, x0+nx) x [y0, y0+ny): one thread per pixel and 4 channels, literal arithmetic
// (conv zero padding outside the net input, u8 zeros on the letterbox canvas), w27 in the graph's RGB order.
__global__ __launch_bounds__(256) void stem_conv_border_kernel(const uint8_t* __restrict__ src, long img_stride, int srcH, int srcW, int step, int inH,
                                                               int inW, int stride, int Ho, int Wo, int Cout, int x0, int y0, int nx, int ny, int B,
                                                               const float* __restrict__ w27, const float* __restrict__ bias,
                                                               const float* __restrict__ slope, int act, float* __restrict__ out1,
                                                               float* __restrict__ out2, const float* __restrict__ s2, const float* __restrict__ t2) {
    const int G = Cout >> 2;
    const int top = y0 * Wo, bottom = (Ho - y0 - ny) * Wo, side = Wo - nx;      // pixels above / below the interior rows, and per interior row beside it
    const int per_img = top + bottom + ny * side;
    const long total = (long)B * per_img * G;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % G);
        const long pi = t / G;
        const int b = (int)(pi / per_img);
        int q = (int)(pi - (long)b * per_img), ox, oy;
        if (q < top) { oy = q / Wo; ox = q - oy * Wo; }
        else if (q < top + bottom) { q -= top; oy = y0 + ny + q / Wo; ox = q % Wo; }
        else { q -= top + bottom; oy = y0 + q / side; const int j = q % side; ox = j < x0 ? j : j + nx; }
        const uint8_t* img = src + (size_t)b * img_stride;
        v4f acc = *reinterpret_cast<const v4f*>(bias + c4 * 4);
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * stride - 1 + ky, ix = ox * stride - 1 + kx;
                if ((unsigned)iy >= (unsigned)inH || (unsigned)ix >= (unsigned)inW) continue;     // conv zero padding
                float vb = 0.f, vg = 0.f, vr = 0.f;                                                  // letterbox canvas = u8 zeros
                if (iy < srcH && ix < srcW) {
                    const uint8_t* px = img + (size_t)iy * step + (size_t)ix * 3;
                    vb = (float)px[0]; vg = (float)px[1]; vr = (float)px[2];
                }
                const float x3[3] = {(vr - 127.5f) / 128.0f, (vg - 127.5f) / 128.0f, (vb - 127.5f) / 128.0f};
                for (int ci = 0; ci < 3; ++ci) acc += *reinterpret_cast<const v4f*>(w27 + (size_t)((ky * 3 + kx) * 3 + ci) * Cout + c4 * 4) * x3[ci];
            }
        v4f sl = {0.f, 0.f, 0.f, 0.f};
        if (slope) sl = *reinterpret_cast<const v4f*>(slope + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = act1(acc[e], act, sl[e]);
        const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * Cout + c4 * 4;
        if (out1) *reinterpret_cast<v4f*>(out1 + o) = acc;
        if (out2) *reinterpret_cast<v4f*>(out2 + o) = acc * *reinterpret_cast<const v4f*>(s2 + c4 * 4) + *reinterpret_cast<const v4f*>(t2 + c4 * 4);
    }
}

template <int STRIDE, int COUT>
static void launch_stem_px(const uint8_t* src, long img_stride, int srcH, int srcW, int step, int B, int inH, int inW, const float* w27,
                           const float* bias, const float* wf, const float* biasf, const float* slope, int act, float* out1, float* out2,
                           const float* s2, const float* t2, hipStream_t s) {
    const int Ho = (inH + 2 - 3) / STRIDE + 1, Wo = (inW + 2 - 3) / STRIDE + 1;
    // interior: every tap a real pixel of the pasted image, and each row's 12-byte window inside that row (not its first / last pixel)
    int x0 = (2 + STRIDE - 1) / STRIDE, x1 = std::min(Wo, (srcW - 3) / STRIDE + 1), y0 = 1, y1 = std::min(Ho, (srcH - 2) / STRIDE + 1);
    if (srcW < 4 || srcH < 3 || x1 <= x0 || y1 <= y0) { x0 = y0 = 0; x1 = y1 = 0; }
    const int nx = x1 - x0, ny = y1 - y0;
    if ((long)B * ny * nx >= (1L << 31)) throw std::runtime_error("stem conv: batch too large for the 32-bit pixel index (split the batch)");
    if (nx > 0)
        hipLaunchKernelGGL((stem_conv_px_kernel<STRIDE, COUT>), dim3((unsigned)(((long)B * ny * nx + 255) / 256)), dim3(256), 0, s, src, img_stride, step, Ho,
                           Wo, x0, y0, nx, ny, B, wf, biasf, slope, act, out1, out2, s2, t2);
    const long frame = (long)B * ((long)Ho * Wo - (long)ny * nx) * (COUT / 4);
    if (frame > 0)
        hipLaunchKernelGGL(stem_conv_border_kernel, dim3(grid_for(frame)), dim3(256), 0, s, src, img_stride, srcH, srcW, step, inH, inW, STRIDE, Ho, Wo,
                           COUT, x0, y0, nx, ny, B, w27, bias, slope, act, out1, out2, s2, t2);
}

static bool stem_px_enabled() {                               // tuning hook (A/B): FACEHIP_STEM_PX=0 -> the LDS-tile kernel for every shape
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_STEM_PX"); v = e ? atoi(e) : 1; }
    return v != 0;
}


// ---- matrix-core form for stems of 16..64 output channels (IResNet: 64) ------------------------------------------------------------
// The LDS-tile kernel above is bound by its 27 x Cout scalar-operand FMAs and LDS reads (155 us for 128 faces, 411 MB written: 0.33 of
// HBM).  Same construction as front_kernel's stem (dwpw_mfma.hip): per 16 output pixels one v_mfma_f32_16x16x32_bf16 shape per 16
// output channels — rows = channels (A = weights as three bf16 terms whose sum is the fp32 weight exactly), columns = pixels (B = the 27
// u8 taps of the 3 x 9-byte window, exact in bf16, 127.5 for the conv padding), fp32 accumulate; the result of the MFMA is pixel x 4
// consecutive channels = one 16-byte store.  Persistent workgroups, a tile = 16 x 16 output pixels, its u8 window prefetched one tile
// ahead into registers and parked in LDS, one LDS-only barrier per tile; per-channel vectors (bias, slope, s2, t2) in LDS so that no
// global load sits between the stores.  Border tiles run a second, per-byte pass for the pixels whose window leaves the frame.
__device__ __forceinline__ void stem_barrier() {                    // workgroup barrier that orders LDS traffic only (no vmcnt drain)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
typedef unsigned stem_v4u __attribute__((ext_vector_type(4)));
typedef __bf16 stem_bf16x8 __attribute__((ext_vector_type(8)));

template <int STRIDE, int CB>
__global__ __launch_bounds__(256, 3) void stem_mfma_kernel(const uint8_t* __restrict__ src, long img_stride, int srcH, int srcW, int step, int inH,
                                                           int inW, int Ho, int Wo, const unsigned* __restrict__ wfrag, const float* __restrict__ biasf,
                                                           const float* __restrict__ slope, int act, float* __restrict__ out1,
                                                           float* __restrict__ out2, const float* __restrict__ s2, const float* __restrict__ t2,
                                                           int tiles_x, int tiles_y, int tiles_total) {
    constexpr int S = STRIDE, T = STEM_TILE, COUT = CB * 16;
    constexpr int ROWS = (T - 1) * S + 3;                                  // staged window rows
    constexpr int PITCH = S == 1 ? 16 : 32;                                // dwords per staged row: ((T-1)*S + 3) * 3 + 3 bytes
    constexpr int SLOTS = (ROWS * PITCH + 255) / 256;
    static_assert(((T - 1) * S + 3) * 3 + 3 <= PITCH * 4, "staged row too narrow");
    __shared__ unsigned stage[2][ROWS * PITCH];
    __shared__ float cst[4][COUT];                                          // bias (folded), slope, s2, t2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lp = lane & 15;
    const int row_bytes = srcW * 3;
    if (tid < COUT) {
        cst[0][tid] = biasf[tid];
        cst[1][tid] = slope ? slope[tid] : 0.f;
        cst[2][tid] = out2 ? s2[tid] : 0.f;
        cst[3][tid] = out2 ? t2[tid] : 0.f;
    }
    stem_v4u wq[CB][3];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int q = 0; q < 3; ++q) wq[cb][q] = reinterpret_cast<const stem_v4u*>(wfrag)[(cb * 3 + q) * 64 + lane];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) asm volatile("" ::"v"(wq[cb][0]), "v"(wq[cb][1]), "v"(wq[cb][2]));   // (the compiler's own wait for these loads: here)
#endif
    // a pixel's window lies inside the frame iff  AY0 <= oy <= AY1  and  AX0 <= ox <= AX1
    const int AY0 = 1, AY1 = min(Ho - 1, (srcH - 2) / S), AX0 = (2 + S - 1) / S, AX1 = min(Wo - 1, (srcW - 3) / S);

    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, wgs = gridDim.x >> 3;      // (gridDim.x is a multiple of 8)
    const int q8 = tiles_total >> 3, r8 = tiles_total & 7;
    const int run0 = xcd * q8 + min(xcd, r8), run1 = run0 + q8 + (xcd < r8 ? 1 : 0);
    const int d_tx = wgs % tiles_x, d_ty = (wgs / tiles_x) % tiles_y, d_n = wgs / (tiles_x * tiles_y);
    int t = run0 + wg;
    int n = t / (tiles_x * tiles_y), tyi = (t / tiles_x) % tiles_y, txi = t % tiles_x;      // the tile being PREFETCHED
    auto advance = [&]() {
        txi += d_tx; if (txi >= tiles_x) { txi -= tiles_x; ++tyi; }
        tyi += d_ty; if (tyi >= tiles_y) { tyi -= tiles_y; ++n; }
        n += d_n;
    };
    unsigned pf[SLOTS];
    const int pr = tid / PITCH, pd4 = (tid % PITCH) * 4;
    auto prefetch = [&]() __attribute__((always_inline)) {
        const uint8_t* frame = src + (size_t)n * img_stride;
        const int sy0 = tyi * T * S - 1, sx3 = (txi * T * S - 1) * 3;
        const unsigned base_lo = (unsigned)(unsigned long long)frame + (unsigned)(sy0 * step + sx3);
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int r = pr + (256 / PITCH) * k;
            const unsigned shr = (base_lo + (unsigned)(r * step)) & 3u;
            const int rel = sx3 - (int)shr + pd4;
            const bool ok = r < ROWS && (unsigned)(sy0 + r) < (unsigned)srcH && rel >= 0 && rel + 4 <= row_bytes;
            pf[k] = ok ? *(const gmem_u32*)(frame + (unsigned)((sy0 + r) * step + rel)) : 0u;
        }
    };
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    stem_barrier();
    if (t < run1) prefetch();
    int buf = 0;
    for (; t < run1; t += wgs, buf ^= 1) {
        const int cn = n, ty0 = tyi * T, tx0 = txi * T;
        const uint8_t* frame = src + (size_t)cn * img_stride;
        const int sy0 = ty0 * S - 1, sx0 = tx0 * S - 1;
        const unsigned base_lo = (unsigned)(unsigned long long)frame + (unsigned)(sy0 * step + sx0 * 3);
#pragma unroll
        for (int k = 0; k < SLOTS; ++k)
            if (tid + 256 * k < ROWS * PITCH) stage[buf][tid + 256 * k] = pf[k];
        advance();
        if (t + wgs < run1) prefetch();
        stem_barrier();
        const unsigned* st = stage[buf];
        const bool tile_inside = ty0 >= AY0 && ty0 + T - 1 <= AY1 && tx0 >= AX0 && tx0 + T - 1 <= AX1;
        auto finish = [&](const float (&f)[8], bool store, int oy, int ox) __attribute__((always_inline)) {
            stem_v4u bq;                                                   // fp32 -> bf16 by truncation: exact for 0..255 and 127.5
#pragma unroll
            for (int i = 0; i < 4; ++i)
                bq[i] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, f[2 * i + 1]), __builtin_bit_cast(unsigned, f[2 * i]), 0x07060302u);
            const stem_bf16x8 bfrag = __builtin_bit_cast(stem_bf16x8, bq);
            const size_t o = (((size_t)cn * Ho + oy) * Wo + ox) * COUT + 4 * g;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                v4f a4 = *reinterpret_cast<const v4f*>(&cst[0][cb * 16 + 4 * g]);
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(stem_bf16x8, wq[cb][2]), bfrag, a4, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(stem_bf16x8, wq[cb][1]), bfrag, a4, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(stem_bf16x8, wq[cb][0]), bfrag, a4, 0, 0, 0);
                if (act == 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a4[c] = a4[c] > 0.f ? a4[c] : 0.f;
                } else if (act == 2) {
                    const v4f sl = *reinterpret_cast<const v4f*>(&cst[1][cb * 16 + 4 * g]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) a4[c] = a4[c] >= 0.f ? a4[c] : a4[c] * sl[c];
                } else if (act == 3) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a4[c] = 1.0f / (1.0f + expf(-a4[c]));
                }
                if (store) {
                    if (out1) *reinterpret_cast<v4f*>(out1 + o + cb * 16) = a4;
                    if (out2) {
                        const v4f sc = *reinterpret_cast<const v4f*>(&cst[2][cb * 16 + 4 * g]), sh = *reinterpret_cast<const v4f*>(&cst[3][cb * 16 + 4 * g]);
                        *reinterpret_cast<v4f*>(out2 + o + cb * 16) = a4 * sc + sh;
                    }
                }
            }
        };
        // FAST pass: wave w takes the tile rows w, w + 4, ...; lane = pixel lp of the row, K block g
#pragma unroll
        for (int i = 0; i < T / 4; ++i) {
            const int py = wid + 4 * i;
            const int oy = ty0 + py, ox = tx0 + lp;
            const bool live = oy < Ho && ox < Wo;
            const bool fast = live && (tile_inside || (oy >= AY0 && oy <= AY1 && ox >= AX0 && ox <= AX1));
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
            if (fast) {
                const int cb3 = lp * S * 3;                                 // window's first byte relative to the staged row's first byte
                if (g < 3) {
                    const int rr = py * S + g;
                    const unsigned pos = ((base_lo + (unsigned)(rr * step)) & 3u) + (unsigned)cb3;
                    const unsigned* q = st + rr * PITCH + (pos >> 2);
                    const unsigned sh = pos & 3u;
                    const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
                    const unsigned n0 = __builtin_amdgcn_alignbyte(d1, d0, sh), n1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    f[0] = (float)(n0 & 255u); f[1] = (float)((n0 >> 8) & 255u); f[2] = (float)((n0 >> 16) & 255u); f[3] = (float)(n0 >> 24);
                    f[4] = (float)(n1 & 255u); f[5] = (float)((n1 >> 8) & 255u); f[6] = (float)((n1 >> 16) & 255u); f[7] = (float)(n1 >> 24);
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int rr = py * S + r;
                        const unsigned pos = ((base_lo + (unsigned)(rr * step)) & 3u) + (unsigned)(cb3 + 8);
                        f[r] = (float)((st[rr * PITCH + (pos >> 2)] >> (8u * (pos & 3u))) & 255u);
                    }
                }
            }
            finish(f, fast, oy, ox);
        }
        if (!tile_inside) {
            // BORDER pass: live pixels the FAST pass skipped — per byte; outside the net input = conv zero padding (127.5 cancels against
            // the folded bias), inside it but outside the pasted image = letterbox canvas (u8 0)
#pragma unroll 1
            for (int i = 0; i < T / 4; ++i) {
                const int py = wid + 4 * i;
                const int oy = ty0 + py, ox = tx0 + lp;
                const bool live = oy < Ho && ox < Wo;
                const bool slow = live && !(oy >= AY0 && oy <= AY1 && ox >= AX0 && ox <= AX1);
                const int iy0 = oy * S - 1, ix0 = ox * S - 1;
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = 0.f;
                if (slow) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = g < 3 ? g : j, rb = g < 3 ? j : 8;
                        if (g == 3 && j >= 3) continue;
                        const int iy = iy0 + r, ix = ix0 + rb / 3;
                        float b = 127.5f;
                        if ((unsigned)iy < (unsigned)inH && (unsigned)ix < (unsigned)inW) {
                            b = 0.f;
                            if (iy < srcH && ix < srcW) b = (float)frame[(size_t)iy * step + (size_t)ix * 3 + rb % 3];
                        }
                        f[j] = b;
                    }
                }
                finish(f, slow, oy, ox);
            }
        }
    }
}

template <int STRIDE>
static bool launch_stem_mfma(const uint8_t* src, long img_stride, int srcH, int srcW, int step, int B, int inH, int inW, int Cout,
                             const unsigned* wfrag, const float* biasf, const float* slope, int act, float* out1, float* out2, const float* s2,
                             const float* t2, hipStream_t s) {
    const int Ho = (inH + 2 - 3) / STRIDE + 1, Wo = (inW + 2 - 3) / STRIDE + 1;
    const int tiles_x = (Wo + STEM_TILE - 1) / STEM_TILE, tiles_y = (Ho + STEM_TILE - 1) / STEM_TILE;
    const int tiles_total = B * tiles_x * tiles_y;
    if ((long)srcH * step >= (1L << 31)) return false;
    const int grid = std::max(8, std::min((tiles_total + 7) / 8 * 8, conv_num_cus() * 3) / 8 * 8);
#define FH_STEM_MFMA(CB)                                                                                                                   \
    hipLaunchKernelGGL((stem_mfma_kernel<STRIDE, CB>), dim3(grid), dim3(256), 0, s, src, img_stride, srcH, srcW, step, inH, inW, Ho, Wo, wfrag, \
                       biasf, slope, act, out1, out2, s2, t2, tiles_x, tiles_y, tiles_total)
    switch (Cout) {
        case 48: FH_STEM_MFMA(3); return true;
        case 64: FH_STEM_MFMA(4); return true;
        default: return false;
    }
#undef FH_STEM_MFMA
}

void launch_stem_conv_u8(const uint8_t* src, long img_stride, int srcH, int srcW, int step, int B, int inH, int inW, int stride,
                         int Cout, const float* w27, const float* bias, const float* wf, const float* biasf, const float* slope, int act,
                         float* out1, float* out2, const float* s2, const float* t2, hipStream_t s, const unsigned* wfrag) {
    static const bool mfma_on = !(getenv("FACEHIP_STEM_MFMA") && atoi(getenv("FACEHIP_STEM_MFMA")) == 0);       // A/B switch
    if (wfrag && biasf && mfma_on && Cout > 32) {                  // (16 / 32 channels: the thread-per-pixel kernel below is at its output-write bound)
        if (stride == 1 ? launch_stem_mfma<1>(src, img_stride, srcH, srcW, step, B, inH, inW, Cout, wfrag, biasf, slope, act, out1, out2, s2, t2, s)
                        : launch_stem_mfma<2>(src, img_stride, srcH, srcW, step, B, inH, inW, Cout, wfrag, biasf, slope, act, out1, out2, s2, t2, s))
            return;
    }
    if (wf && stem_px_enabled()) {
#define FH_STEM_PX(S, C) launch_stem_px<S, C>(src, img_stride, srcH, srcW, step, B, inH, inW, w27, bias, wf, biasf, slope, act, out1, out2, s2, t2, s); return
        if (stride == 2 && Cout == 16) { FH_STEM_PX(2, 16); }
        if (stride == 2 && Cout == 32) { FH_STEM_PX(2, 32); }
        if (stride == 1 && Cout == 16) { FH_STEM_PX(1, 16); }
        if (stride == 1 && Cout == 32) { FH_STEM_PX(1, 32); }
        // (Cout = 64, IResNet's stem: 64 accumulators + 108 scalar loads per thread — measured 545 us against 176 us for the LDS-tile
        //  kernel below at B = 128, whose 4-channels-per-thread layout also writes whole 256-byte pixel rows per wave)
#undef FH_STEM_PX
    }
    const int Ho = (inH + 2 - 3) / stride + 1, Wo = (inW + 2 - 3) / stride + 1;
    const int ntiles = B * ((Wo + STEM_TILE - 1) / STEM_TILE) * ((Ho + STEM_TILE - 1) / STEM_TILE);
    const int blocks = ntiles < 256 * 8 ? ntiles : 256 * 8;
    if (stride == 1)
        hipLaunchKernelGGL(stem_conv_u8_kernel<1>, dim3(blocks), dim3(256), 0, s, src, img_stride, srcH, srcW, step, inH, inW, Ho, Wo, Cout,
                           w27, bias, slope, act, out1, out2, s2, t2, ntiles);
    else
        hipLaunchKernelGGL(stem_conv_u8_kernel<2>, dim3(blocks), dim3(256), 0, s, src, img_stride, srcH, srcW, step, inH, inW, Ho, Wo, Cout,
                           w27, bias, slope, act, out1, out2, s2, t2, ntiles);
}


}  // namespace fh
