// conv_halo.hip — dense 3x3 stride-1 convolutions with 16 input channels on large maps: SCRFD's FPN smoothing convolutions (16 -> 16),
// executed by ORT inside session_->Run (reference src/face_detector.cpp:179-183).
//
// As an implicit GEMM these layers re-read every input pixel nine times from L2 (K = 144 is short, N = 16 is narrow) and
// conv_igemm_kernel runs them at 14-37 TFLOP/s.  Measured at B = 128 (det_500m): 80x80 101 -> 77 us, 40x40 33.5 -> 28 us.  The same kernel
// was measured for the 64 -> 30 head convolutions (template CIN4 = 16) and is NOT used there: 315 against 319 us at 80x80, slower on
// the 40x40 / 20x20 maps — those layers are bound by the matrix cores (N padded 30 -> 32), not by their input traffic.  Here one workgroup
// owns an 8 x 16 SPATIAL tile: its 10 x 18 input halo goes global -> LDS once (LDS-DMA, channels-last, zero line for the padding,
// 16-byte columns XOR-swizzled by the halo column so that the fragment reads of 16 neighbouring pixels spread over all banks), and the
// nine taps are nine shifted views of that one LDS image — the MFMA pixel fragments (v_mfma_f32_32x32x2_f32, B operand) are ds_read_b128
// at (py + ky, px + kx).  The weights are the A operand: pre-packed at load time in fragment order [tap][k-step][n-tile][lane], each wave
// fetches them straight into registers (coalesced 1 KB loads, L1 / L2 resident: every workgroup reads the same <= 72 KB), so there is
// no second LDS image and exactly ONE barrier per tile.  Epilogue as conv_mfma.hip: bias -> ReLU / sigmoid -> + residual -> store, or the
// per-channel-range outputs of merged sibling convolutions.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernels.h"
#include "plan.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void halo_dma16(const float* src, v4f* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

constexpr int CH_TH = 8, CH_TW = 16, CH_HW = CH_TW + 2, CH_HALO = (CH_TH + 2) * CH_HW;      // 128 outputs, 10 x 18 = 180 halo pixels

__device__ __forceinline__ float halo_act(float v, int act) {
    if (act == (int)Act::RELU) return v > 0.f ? v : 0.f;
    if (act == (int)Act::SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// CIN4 = float4 columns per pixel (4: Cin = 16, 16: Cin = 64); TN = 32-wide output-channel tiles (Cout <= 32 * TN)
template <int CIN4, int TN>
__global__ __launch_bounds__(256, CIN4 == 4 ? 4 : 3) void conv3x3_halo_kernel(const ConvArgs p, const v4f* __restrict__ wfrag, const int tiles_x,
                                                                             const int tiles_y) {
    constexpr int SLOTS = (CH_HALO * CIN4 + 255) / 256 * 256;
    constexpr int KS = CIN4 / 2;                                   // 8-deep k-steps per tap
    __shared__ v4f halo[SLOTS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh2 = lane >> 5;
    const int C = p.Cin;
    int t;
    {   // XCD-contiguous tile order: neighbouring tiles (shared halo rows / columns) on one L2
        const int nb = gridDim.x, q = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        t = x * q + min(x, r8) + (int)(blockIdx.x >> 3);
    }
    const int tx0 = (t % tiles_x) * CH_TW; t /= tiles_x;
    const int ty0 = (t % tiles_y) * CH_TH;
    const int n = t / tiles_y;
    const float* img = p.in + (size_t)n * p.H * p.W * C;
    auto key = [](int hx) { return CIN4 == 16 ? (hx & 15) : ((hx >> 2) & 3); };

    // ---- halo: slot s = j*256 + tid holds halo pixel s / CIN4, physical column s % CIN4 = logical column ^ key(hx)
#pragma unroll
    for (int j = 0; j < SLOTS / 256; ++j) {
        const int s = j * 256 + tid, hp = s / CIN4, cp = s % CIN4;
        const float* src = p.zeros;
        if (hp < CH_HALO) {
            const int hy = hp / CH_HW, hx = hp - hy * CH_HW;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) src = img + ((size_t)iy * p.W + ix) * C + ((cp ^ key(hx)) * 4);
        }
        halo_dma16(src, halo + j * 256 + wid * 64);
    }
    v16f acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int pix = wid * 32 + fr, py = pix >> 4, px = pix & 15;
    const v4f* wl = wfrag + lane;
    // weights: global -> registers, always at least one tap ahead of the MFMAs that use them (every wave reads the same L1 / L2-resident
    // image; left next to their use, each 1 KB load costs its full latency against only 4 MFMAs).  Cin = 16 holds all nine taps at once
    // — issued before the halo barrier, so they arrive while the halo does; Cin = 64 double-buffers one tap (8 k-steps x TN float4).
    constexpr bool WALL = 9 * KS * TN <= 36;
    constexpr int WT = KS * TN;                                    // float4 fragments per tap
    v4f wa[WALL ? 9 * WT : WT], wb[WALL ? 1 : WT];
    if (WALL) {
#pragma unroll
        for (int i = 0; i < 9 * WT; ++i) wa[i] = wl[i * 64];
    } else {
#pragma unroll
        for (int i = 0; i < WT; ++i) wa[i] = wl[i * 64];
    }
    __syncthreads();                                               // (drains vmcnt: the halo has landed)
    auto tap_mma = [&](int tap, const v4f* w) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int hx = px + kx;
        const v4f* hrow = halo + ((py + ky) * CH_HW + hx) * CIN4;
        const int kk = key(hx);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4f x = hrow[(2 * s + fh2) ^ kk];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s * TN + j][e], x[e], acc[j], 0, 0, 0);
        }
    };
    if (WALL) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) tap_mma(tap, wa + tap * WT);
    } else {
#pragma unroll
        for (int tap = 0; tap < 9; tap += 2) {
            if (tap + 1 < 9) {
#pragma unroll
                for (int i = 0; i < WT; ++i) wb[i] = wl[((tap + 1) * WT + i) * 64];
            }
            tap_mma(tap, wa);
            if (tap + 1 < 9) {
                if (tap + 2 < 9) {
#pragma unroll
                    for (int i = 0; i < WT; ++i) wa[i] = wl[((tap + 2) * WT + i) * 64];
                }
                tap_mma(tap + 1, wb);
            }
        }
    }
    // ---- epilogue: lane = pixel, accumulator quads = 4 consecutive channels  (C/D map: row = (e&3) + 8*(e>>2) + 4*(lane>>5))
    const int oy = ty0 + py, ox = tx0 + px;
    if (oy >= p.Ho || ox >= p.Wo) return;
    const size_t m = ((size_t)n * p.Ho + oy) * p.Wo + ox;
    if (p.n_outs > 0) {                                            // merged sibling convs: per-channel-range destination
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = j * 32 + 4 * fh2 + 8 * (e >> 2) + (e & 3);
                if (co >= p.Cout) continue;
                const int g = co >= p.oc0[2] && p.n_outs > 2 ? 2 : co >= p.oc0[1] ? 1 : 0;
                const int cg = p.oc0[g + 1] - p.oc0[g];
                p.outs[g][m * cg + (co - p.oc0[g])] = halo_act(acc[j][e] + p.bias[co], p.oact[g]);
            }
        return;
    }
    const bool vec = (p.Cout & 3) == 0;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = j * 32 + 4 * fh2 + 8 * g;
            if (co >= p.Cout) continue;
            if (vec) {
                const v4f b4 = *reinterpret_cast<const v4f*>(p.bias + co);
                v4f v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = halo_act(acc[j][4 * g + c] + b4[c], p.act);
                if (p.res) v += *reinterpret_cast<const v4f*>(p.res + m * p.Cout + co);
                *reinterpret_cast<v4f*>(p.out1 + m * p.Cout + co) = v;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (co + c >= p.Cout) continue;
                    float v = halo_act(acc[j][4 * g + c] + p.bias[co + c], p.act);
                    if (p.res) v += p.res[m * p.Cout + co + c];
                    p.out1[m * p.Cout + co + c] = v;
                }
            }
        }
}

// ---- Cout <= 16 (SCRFD's FPN smoothing convolutions, 16 -> 16): v_mfma_f32_16x16x4_f32 instead of padding 16 output channels to the
// 32 rows of the 32x32 instruction (half of the kernel's matrix-core time multiplied zeros: 80x80 at B = 128 needs 55 us of 32x32x2 issue
// for 77 us of kernel).  Same halo image, same one barrier.  A wave owns two 16-pixel row segments; lane (j = lane & 15, q = lane >> 4)
// reads ONE float4 per tap — channels 4q .. 4q+3 of pixel j under that tap — and feeds element s of it to the s-th of four MFMAs: the K
// index k of step s stands for channel 4k + s, and the weight lane (cout i, k = q) holds the matching float4 w[i][tap][4q .. 4q+3], the
// filter's natural layout.  D comes back as pixel j, output channels 4q .. 4q+3: one float4 store per lane and segment.
static inline bool halo16_shape(int Cin, int Cout, int n_outs) { return Cin == 16 && Cout <= 16 && (Cout & 3) == 0 && n_outs == 0; }

__global__ __launch_bounds__(256, 4) void conv3x3_halo16_kernel(const ConvArgs p, const v4f* __restrict__ wfrag, const int tiles_x, const int tiles_y) {
    constexpr int SLOTS = (CH_HALO * 4 + 255) / 256 * 256;
    __shared__ v4f halo[SLOTS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    int t;
    {
        const int nb = gridDim.x, qq = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        t = x * qq + min(x, r8) + (int)(blockIdx.x >> 3);
    }
    const int tx0 = (t % tiles_x) * CH_TW; t /= tiles_x;
    const int ty0 = (t % tiles_y) * CH_TH;
    const int n = t / tiles_y;
    const float* img = p.in + (size_t)n * p.H * p.W * 16;
#pragma unroll
    for (int jj = 0; jj < SLOTS / 256; ++jj) {                      // halo: as conv3x3_halo_kernel (physical column = logical ^ key(hx))
        const int s = jj * 256 + tid, hp = s >> 2, cp = s & 3;
        const float* src = p.zeros;
        if (hp < CH_HALO) {
            const int hy = hp / CH_HW, hx = hp - hy * CH_HW;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) src = img + ((size_t)iy * p.W + ix) * 16 + ((cp ^ ((hx >> 2) & 3)) * 4);
        }
        halo_dma16(src, halo + jj * 256 + wid * 64);
    }
    v4f w[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) w[tap] = wfrag[tap * 64 + lane];  // (issued before the barrier: they arrive while the halo does)
    v4f acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();                                               // (drains vmcnt: the halo has landed)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int hx = j + kx;
        const int col = q ^ ((hx >> 2) & 3);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const v4f x = halo[((2 * wid + g + ky) * CH_HW + hx) * 4 + col];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][s], x[s], acc[g], 0, 0, 0);
        }
    }
    const int co = 4 * q;
    if (co >= p.Cout) return;
    const v4f b4 = *reinterpret_cast<const v4f*>(p.bias + co);
    const int ox = tx0 + j;
    v4f r4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    size_t m[2]; bool ok[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {                                   // residual reads before the stores (loads and stores share vmcnt)
        const int oy = ty0 + 2 * wid + g;
        ok[g] = oy < p.Ho && ox < p.Wo;
        m[g] = ((size_t)n * p.Ho + min(oy, p.Ho - 1)) * p.Wo + min(ox, p.Wo - 1);
        if (p.res) r4[g] = *reinterpret_cast<const v4f*>(p.res + m[g] * p.Cout + co);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        v4f v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = halo_act(acc[g][c] + b4[c], p.act);
        v += r4[g];
        if (ok[g]) *reinterpret_cast<v4f*>(p.out1 + m[g] * p.Cout + co) = v;
    }
}

// Can this convolution take the spatial-tile kernel?  (3x3 stride 1 pad 1, Cin = 16, Cout <= 64, ReLU / sigmoid / none, residual of
// the same shape or none, no second output; maps large enough that 8 x 16 tiles are mostly full)
bool conv_halo_ok(const ConvArgs& a) {
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 16 && a.Cout <= 64 && a.act != (int)Act::PRELU && !a.out2 &&
           (a.res_mode == (int)ResMode::NONE || a.res_mode == (int)ResMode::SAME) && a.H >= 16 && a.W >= 16 && a.dw_w == nullptr && !a.bias_cls &&
           !(a.n_outs > 0 && a.Cout <= 16 && (a.Cout & 3) == 0);        // (that shape's weights are packed for the 16x16x4 kernel, which has no merged outputs)
}

size_t conv_halo_wfrag_floats(int Cin, int Cout) { return (size_t)9 * (Cin / 8) * ((Cout + 31) / 32) * 64 * 4; }   // (the 16x16x4 form uses the first half)

// host: plan-layout weights [Cout][9][Cin] -> fragment order [tap][k-step s][n-tile j][lane][4]: lane (fr, fh2) holds output channel
// j*32 + fr, input channels (2s + fh2)*4 .. +3 of that tap (rows >= Cout are zero)
void conv_halo_pack_weights(const float* w, int Cout, int Cin, float* dst) {
    if (Cin == 16 && Cout <= 16 && (Cout & 3) == 0) {
        // conv3x3_halo16_kernel: [tap][lane][4], lane (i = lane & 15, q = lane >> 4) = w[cout i][tap][4q .. 4q+3] (rows >= Cout zero).  A
        // merged-output convolution of this shape (n_outs > 0) would need the 32x32 layout: conv_halo_ok refuses it.
        for (int tap = 0; tap < 9; ++tap)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = lane & 15, ci = (lane >> 4) * 4;
                float* d = dst + ((size_t)tap * 64 + lane) * 4;
                for (int c = 0; c < 4; ++c) d[c] = co < Cout ? w[((size_t)co * 9 + tap) * Cin + ci + c] : 0.f;
            }
        return;
    }
    const int KS = Cin / 8, TN = (Cout + 31) / 32;
    for (int tap = 0; tap < 9; ++tap)
        for (int s = 0; s < KS; ++s)
            for (int j = 0; j < TN; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = j * 32 + (lane & 31), ci = (2 * s + (lane >> 5)) * 4;
                    float* d = dst + ((((size_t)tap * KS + s) * TN + j) * 64 + lane) * 4;
                    for (int c = 0; c < 4; ++c) d[c] = co < Cout ? w[((size_t)co * 9 + tap) * Cin + ci + c] : 0.f;
                }
}

void launch_conv_halo(const ConvArgs& a0, const float* wfrag, hipStream_t s) {
    ConvArgs a = a0;
    if (!conv_halo_ok(a)) throw std::runtime_error("conv_halo: unsupported convolution");
    if ((long)a.B * a.H * a.W * a.Cin >= (1L << 31) || (long)a.B * a.Ho * a.Wo * 64 >= (1L << 31))
        throw std::runtime_error("conv_halo: tensor too large for one launch (split the batch)");
    a.zeros = conv_zero_line();
    const int tiles_x = (a.Wo + CH_TW - 1) / CH_TW, tiles_y = (a.Ho + CH_TH - 1) / CH_TH;
    const dim3 grid((unsigned)(a.B * tiles_y * tiles_x));
    const v4f* wf = reinterpret_cast<const v4f*>(wfrag);
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    if (halo16_shape(a.Cin, a.Cout, a.n_outs)) hipLaunchKernelGGL(conv3x3_halo16_kernel, grid, dim3(256), 0, s, a, wf, tiles_x, tiles_y);
    else if (a.Cout <= 32) hipLaunchKernelGGL((conv3x3_halo_kernel<4, 1>), grid, dim3(256), 0, s, a, wf, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<4, 2>), grid, dim3(256), 0, s, a, wf, tiles_x, tiles_y);
    timer.end(s, 9, a.t_flops, a.t_bytes);
}

}  // namespace fh
