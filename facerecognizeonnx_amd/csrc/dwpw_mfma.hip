// dwpw_mfma.hip — depthwise 3x3 of stride 1 or 2 (+bias +ReLU) fused with the pointwise 1x1 convolution that
// consumes it: the MobileNet "conv_dw" block of SCRFD (ONNX Conv(group=C) -> Relu -> Conv 1x1 -> Relu,
// executed by ORT inside session_->Run, reference src/face_detector.cpp:179-183).
//
// Run separately, these HBM-bound blocks write the depthwise result and read it straight back —
// as much traffic as the block's real input and output together.  Here one workgroup owns a SPATIAL
// tile of 8 x 16 output pixels and walks the channels in chunks of 32:
//   1. stride 1: the (8+2) x (16+2) input halo of the chunk goes global -> LDS by LDS-DMA (each element once,
//      coalesced 128-byte pixel rows; out-of-image pixels come from the zero line, 16-byte channel columns
//      >= C are not loaded at all), together with the pointwise weight chunk [BN][32];
//      stride 2: no halo — vertical strips of outputs gather their taps from global memory (see DIRECT);
//   2. the depthwise 3x3 (+bias +activation) is evaluated from LDS on the vector ALU, 4 channels per
//      lane, and written to LDS as the GEMM's A tile [128 pixels][32 k] (XOR-swizzled like conv_mfma.hip);
//   3. the pointwise product accumulates on v_mfma_f32_32x32x2_f32 exactly as in conv_igemm_kernel.
// The intermediate tensor never exists in HBM.  Two barriers per chunk, no double buffering: the
// layers are bandwidth-bound and >= 2 workgroups per CU overlap each other's phases.
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "kernels.h"
#include "plan.h"
#include "stem_fma.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dwpw_dma16(const float* src, v4f* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

typedef const unsigned __attribute__((address_space(1))) dwpw_gmem_u32;
constexpr int DP_TH = 8, DP_TW = 16, DP_BM = DP_TH * DP_TW;            // 128 output pixels per tile
constexpr int DP_HW = DP_TW + 2, DP_HALO = (DP_TH + 2) * DP_HW;         // 10 x 18 = 180 halo pixels

// DS = stride of the depthwise part (1 | 2).
// DIRECT = no halo in LDS: a thread owns 4 channels of a vertical strip of SR output pixels and gathers the
// ((SR-1)*DS+3) x 3 input float4s of the strip straight from global memory into registers (re-reads between
// neighbouring strips hit L1 / L2).  The register file is a far larger landing area than LDS (more bytes in
// flight per CU) and the halo barrier disappears; it is the only form for DS == 2, whose 17 x 33 halo would not fit.
// HC = 16-byte channel columns per pixel handled at a time: 8 (a 32-channel chunk), or 4 for layers with C <= 16 —
// half the halo LDS, so more workgroups per CU, and no idle depthwise lanes.
template <int BN, int WM, int WN, int DS, int HC, bool DIRECT, bool STEM = false>
__global__ __launch_bounds__(256, DIRECT ? (HC == 4 ? 3 : BN * DS <= 64 ? 3 : 2) : HC == 4 ? 4 : 2) void dwpw_kernel(const ConvArgs p, const int tiles_x, const int tiles_y, const int tiles_n) {
    static_assert(DS == 1 || DIRECT, "stride 2 needs the direct form");
    static_assert(!STEM || (HC == 4 && !DIRECT && DP_HALO <= 256), "the fused stem produces a 16-channel halo, one pixel per thread");
    constexpr int BM = DP_BM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int HALO_SLOTS = !DIRECT ? (DP_HALO * HC + 255) / 256 * 256 : 0;  // float4 slots, whole DMA passes
    constexpr int PPP = 256 / HC;                                          // pixels per depthwise pass
    constexpr int SR = BM / PPP;                                           // DIRECT: output rows per thread strip
    constexpr int NR = (SR - 1) * DS + 3;                                  //         input rows a strip touches
    __shared__ v4f lds[HALO_SLOTS + BM * 8 + BN * 8];
    v4f* const halo = lds;
    v4f* const At = lds + HALO_SLOTS;
    v4f* const Wt = At + BM * 8;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int C = p.Cin;
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2: neighbouring tiles share halo rows /
    // columns, so give every XCD a CONTIGUOUS run of tiles (whole images, row after row) — the halo is then fetched once per XCD
    // instead of once per tile (rocprofv3, 320x320x16 layer: FETCH_SIZE 1.54x the input with the linear order).
    int t;
    {
        const int nb = gridDim.x, q = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        t = x * q + min(x, r8) + (int)(blockIdx.x >> 3);
    }
    const int tile_n = t % tiles_n; t /= tiles_n;
    const int tx0 = (t % tiles_x) * DP_TW; t /= tiles_x;
    const int ty0 = (t % tiles_y) * DP_TH;
    const int n = t / tiles_y;
    const int n0 = tile_n * BN;
    const int chunks = p.Kpad / 32;
    const float* img = STEM ? nullptr : p.in + (size_t)n * p.H * p.W * C;  // H x W = depthwise input, Ho x Wo = output grid

    // ---- halo loader: slot s = 256*j + tid covers halo pixel s>>3, 16-byte column s&7
    constexpr int HP = !DIRECT ? HALO_SLOTS / 256 : 1;
    long h_off[HP];                                                        // float offset of (pixel, column 0) or -1
#pragma unroll
    for (int j = 0; j < (!DIRECT ? HP : 0); ++j) {
        const int s = j * 256 + tid, hp = s / HC;
        h_off[j] = -1;
        if (hp < DP_HALO) {
            const int hy = hp / DP_HW, hx = hp - hy * DP_HW;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) h_off[j] = ((long)iy * p.W + ix) * C + (s % HC) * 4;
        }
    }
    const int hq = tid % HC;                                               // same for every pass (256 % HC == 0)
    // ---- pointwise weight loader (as conv_mfma.hip: LDS-DMA, swizzle on the source column)
    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);
    constexpr int BL = BN / 32;
    const char* w_base = reinterpret_cast<const char*>(p.wt) + (size_t)n0 * p.Kpad * 4;
    unsigned w_off[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) w_off[i] = (unsigned)(((lrow + i * 32) * p.Kpad + lqs * 4) * 4);

    // ---- depthwise producer: thread -> 16-byte column dq of pixels dp + PPP*i
    const int dq = tid % HC, dp = tid / HC;
    const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int kc = 0; kc < chunks; ++kc) {
        const int c0 = kc * 32;
        const bool hvalid = c0 + hq * 4 < C;                               // 16-byte columns past C are never read: no DMA for them
        const int ksteps = min(4, (C - c0 + 7) >> 3);                      // 8-deep MFMA steps that hold real channels
        __syncthreads();                                                   // previous chunk: halo + fragments fully consumed
        if (STEM) {
            // halo pixel tid of the tile = one output pixel of the stem convolution, computed here from the u8 frame (3 aligned dwords per
            // image row, weights in SGPRs: see stem_conv_px_kernel in ops_misc.hip); pixels outside the map are the depthwise zero padding
            if (tid < DP_HALO) {
                const int hy = tid / DP_HW, hx = tid - hy * DP_HW;
                const int ay = ty0 + hy - 1, ax = tx0 + hx - 1;
                float a16[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) a16[c] = 0.f;
                if ((unsigned)ay < (unsigned)p.H && (unsigned)ax < (unsigned)p.W) {
                    const int S = p.u8_stride;
                    const int iy0 = ay * S - 1, ix0 = ax * S - 1;
                    const uint8_t* frame = p.u8_src + (size_t)n * p.u8_img_stride;
                    float v[27];
                    if (iy0 >= 0 && iy0 + 2 < p.u8_srcH && ix0 >= 1 && ix0 + 2 <= p.u8_srcW - 2) {
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const unsigned long long a = (unsigned long long)(frame + (size_t)(iy0 + r) * p.u8_step + (size_t)ix0 * 3);
                            const unsigned sh = (unsigned)a & 3u;
                            const dwpw_gmem_u32* q = (const dwpw_gmem_u32*)(a - sh);
                            const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
                            const unsigned n0 = __builtin_amdgcn_alignbyte(d1, d0, sh), n1 = __builtin_amdgcn_alignbyte(d2, d1, sh), n2 = d2 >> (8u * sh);
                            v[r * 9 + 0] = (float)(n0 & 255u); v[r * 9 + 1] = (float)((n0 >> 8) & 255u); v[r * 9 + 2] = (float)((n0 >> 16) & 255u); v[r * 9 + 3] = (float)(n0 >> 24);
                            v[r * 9 + 4] = (float)(n1 & 255u); v[r * 9 + 5] = (float)((n1 >> 8) & 255u); v[r * 9 + 6] = (float)((n1 >> 16) & 255u); v[r * 9 + 7] = (float)(n1 >> 24);
                            v[r * 9 + 8] = (float)(n2 & 255u);
                        }
                    } else {
                        // window touches the frame's border: per tap — outside the net input = conv zero padding (127.5 cancels against the
                        // folded bias), inside it but outside the pasted image = letterbox canvas (u8 0)
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const int iy = iy0 + r, ix = ix0 + c;
                                float b0 = 127.5f, b1 = 127.5f, b2 = 127.5f;
                                if ((unsigned)iy < (unsigned)p.u8_inH && (unsigned)ix < (unsigned)p.u8_inW) {
                                    b0 = b1 = b2 = 0.f;
                                    if (iy < p.u8_srcH && ix < p.u8_srcW) {
                                        const uint8_t* px = frame + (size_t)iy * p.u8_step + (size_t)ix * 3;
                                        b0 = (float)px[0]; b1 = (float)px[1]; b2 = (float)px[2];
                                    }
                                }
                                v[r * 9 + c * 3 + 0] = b0; v[r * 9 + c * 3 + 1] = b1; v[r * 9 + c * 3 + 2] = b2;
                            }
                    }
                    fh_v2f a2[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) a2[i] = fh_v2f{p.stem_bf[2 * i], p.stem_bf[2 * i + 1]};
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float* wrow = p.stem_wf + (size_t)(r * 9) * 16;
                        fh_v2f xp[5];
#pragma unroll
                        for (int i = 0; i < 4; ++i) xp[i] = fh_v2f{v[r * 9 + 2 * i], v[r * 9 + 2 * i + 1]};
                        xp[4] = fh_v2f{v[r * 9 + 8], 0.f};
#if defined(__HIP_DEVICE_COMPILE__)
                        FH_STEM_ROW_FMA(a2, xp, wrow, 64);
#else
                        (void)wrow; (void)xp;
#endif
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) { a16[2 * i] = a2[i][0]; a16[2 * i + 1] = a2[i][1]; }
                    if (p.stem_act == (int)Act::RELU) {
#pragma unroll
                        for (int c = 0; c < 16; ++c) a16[c] = a16[c] > 0.f ? a16[c] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) halo[tid * 4 + q] = v4f{a16[4 * q], a16[4 * q + 1], a16[4 * q + 2], a16[4 * q + 3]};
            }
        } else if (!DIRECT && hvalid) {
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                const float* src = h_off[j] >= 0 ? img + h_off[j] + c0 : p.zeros;
                dwpw_dma16(src, halo + j * 256 + wid * 64);
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) dwpw_dma16(reinterpret_cast<const float*>(w_base + w_off[i]), Wt + i * 256 + wid * 64);
        w_base += 128;
        // depthwise weights / bias of this thread's 4 channels (channels >= C: zero line -> zero output)
        const int cw = c0 + dq * 4;
        const bool cvalid = cw < C;
        v4f wk[9], b4;
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const v4f*>(cvalid ? p.dw_w + (size_t)k * C + cw : p.zeros);
        b4 = *reinterpret_cast<const v4f*>(cvalid ? p.dw_b + cw : p.zeros);
        if (!DIRECT) {
            __syncthreads();                                               // halo + weights landed (barrier drains vmcnt)
            if (dq < 2 * ksteps) {
#pragma unroll
                for (int i = 0; i < BM / PPP; ++i) {
                    const int px = dp + PPP * i;
                    const int py = px / DP_TW, pxx = px - py * DP_TW;
                    v4f a = b4;
                    if (cvalid) {                                          // (the invalid half of a partly valid step: zeros, its halo was not loaded)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a += halo[((py + ky) * DP_HW + pxx + kx) * HC + dq] * wk[ky * 3 + kx];
                        if (p.dw_act == (int)Act::RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[e] = a[e] > 0.f ? a[e] : 0.f;
                        }
                    }
                    At[px * 8 + (dq ^ ((px >> 1) & 7))] = a;
                }
            }
        } else {
            if (dq < 2 * ksteps) {
                const int sx = dp % DP_TW, sy = dp / DP_TW;                 // strip: column sx, output rows sy*SR .. sy*SR+SR-1
                const int iy0 = (ty0 + sy * SR) * DS - 1, ix0 = (tx0 + sx) * DS - 1;
                const float* cimg = img + cw;
                v4f a[SR];
#pragma unroll
                for (int o = 0; o < SR; ++o) a[o] = b4;
                if (cvalid) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = ix0 + kx;
                        const bool xin = (unsigned)ix < (unsigned)p.W;
                        v4f x[NR];
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            const int iy = iy0 + r;
                            const bool in = xin && (unsigned)iy < (unsigned)p.H;
                            x[r] = *reinterpret_cast<const v4f*>(in ? cimg + ((long)iy * p.W + ix) * C : p.zeros);
                        }
#pragma unroll
                        for (int o = 0; o < SR; ++o)
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky) a[o] += x[o * DS + ky] * wk[ky * 3 + kx];
                    }
                    if (p.dw_act == (int)Act::RELU) {
#pragma unroll
                        for (int o = 0; o < SR; ++o)
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[o][e] = a[o][e] > 0.f ? a[o][e] : 0.f;
                    }
                }
#pragma unroll
                for (int o = 0; o < SR; ++o) {
                    const int px = (sy * SR + o) * DP_TW + sx;
                    At[px * 8 + (dq ^ ((px >> 1) & 7))] = a[o];
                }
            }
        }
        __syncthreads();                                                   // A tile complete
        const v4f* X = At + (wm * TM * 32 + fr) * 8;
        const v4f* Wp = Wt + (wn * TN * 32 + fr) * 8;
        for (int s = 0; s < ksteps; ++s) {
            const int col = (2 * s + fh2) ^ fsw;
            v4f x[TM], w[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) x[i] = X[i * 32 * 8 + col];
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = Wp[j * 32 * 8 + col];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], x[i][e], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: lane = pixel (lane&31 within each 32-row block), accumulator quads = 4 consecutive channels
    const bool vec = (p.Cout & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + fr;
        const int oy = ty0 + r / DP_TW, ox = tx0 + r % DP_TW;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        float* __restrict__ orow = p.out1 + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cb = n0 + (wn * TN + j) * 32 + 4 * fh2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cb + 8 * g;
                if (co >= p.Cout) continue;
                if (vec) {
                    const v4f bb = *reinterpret_cast<const v4f*>(p.bias + co);
                    v4f v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float u = acc[i][j][4 * g + c] + bb[c];
                        if (p.act == (int)Act::RELU) u = u > 0.f ? u : 0.f;
                        v[c] = u;
                    }
                    *reinterpret_cast<v4f*>(orow + co) = v;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (co + c >= p.Cout) continue;
                        float u = acc[i][j][4 * g + c] + p.bias[co + c];
                        if (p.act == (int)Act::RELU) u = u > 0.f ? u : 0.f;
                        orow[co + c] = u;
                    }
                }
            }
        }
    }
}

template <int BN, int WM, int WN>
static void launch_dwpw_cfg(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.Wo + DP_TW - 1) / DP_TW, tiles_y = (a.Ho + DP_TH - 1) / DP_TH, tiles_n = (a.Cout + BN - 1) / BN;
    const dim3 grid((unsigned)(a.B * tiles_y * tiles_x * tiles_n));
    // (the direct form was measured for stride 1 too: 15-20 % slower than the LDS halo on every SCRFD layer —
    //  L1 traffic of the 4.5-6x re-reads costs more than the extra loads in flight gain)
    if (a.u8_src) {
        if (a.Cin != 16 || a.dw_stride != 1 || BN > 64) throw std::runtime_error("dwpw: the fused stem needs 16 channels, stride 1 and Cout <= 64");
        hipLaunchKernelGGL((dwpw_kernel<(BN <= 64 ? BN : 64), (BN <= 64 ? WM : 2), (BN <= 64 ? WN : 2), 1, 4, false, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
    } else if (a.dw_stride == 2 && a.Cin <= 16 && BN <= 64)     // 16 channels = 4 float4 columns: every depthwise lane busy, 2-row strips (fewer registers)
        hipLaunchKernelGGL((dwpw_kernel<(BN <= 64 ? BN : 64), (BN <= 64 ? WM : 2), (BN <= 64 ? WN : 2), 2, 4, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
    else if (a.dw_stride == 2) hipLaunchKernelGGL((dwpw_kernel<BN, WM, WN, 2, 8, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
    else if (a.Cin <= 16 && BN <= 64) hipLaunchKernelGGL((dwpw_kernel<(BN <= 64 ? BN : 64), (BN <= 64 ? WM : 2), (BN <= 64 ? WN : 2), 1, 4, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
    else hipLaunchKernelGGL((dwpw_kernel<BN, WM, WN, 1, 8, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
}

// a.in = depthwise input [B,H,W,C] (a.dw_stride 1 | 2, pad 1; Ho x Wo = its output grid = the pointwise grid), a.Cin = C, a.wt = packed pointwise weights
// [..][conv_kpad(C)], a.dw_w [9][C], a.dw_b [C]; activation of the pointwise part limited to NONE / RELU.
void launch_dwpw(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.zeros = conv_zero_line();
    if ((long)a.B * a.Ho * a.Wo <= 0) return;
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    if (a.Cout <= 32) launch_dwpw_cfg<32, 4, 1>(a, s);
    else if (a.Cout <= 64) launch_dwpw_cfg<64, 2, 2>(a, s);
    else if (a.Cout <= 96) launch_dwpw_cfg<96, 4, 1>(a, s);
    else launch_dwpw_cfg<128, 2, 2>(a, s);
    timer.end(s, 4, a.t_flops, a.t_bytes);
}

}  // namespace fh
