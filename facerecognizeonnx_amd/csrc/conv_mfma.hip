// conv_mfma.hip — dense convolution (3x3 / 1x1, stride 1|2) and the final FC as an implicit GEMM
// on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (gfx950 / CDNA4).
//
// What it replaces: the Conv / Gemm nodes ONNX Runtime executes inside `session_->Run`
// (reference src/face_recognizer.cpp:279-283, src/face_detector.cpp:179-183), with the
// following BatchNormalization / PRelu / Relu / Sigmoid / Add nodes fused into the epilogue.
//
// GEMM view:  M = B*Ho*Wo pixels, N = Cout, K = ks*ks*Cin with k = tap*Cin + ci.
// Activations are channels-last, so a K-chunk of 32 consecutive k is 128 contiguous bytes of
// one input pixel: every global load is a full 16-byte lane access and 8 lanes cover one line.
//
// Tile anatomy (256 threads = 4 waves, 2 workgroups per CU):
//   * global -> registers -> LDS, double buffered; one barrier per 32-deep K chunk.
//   * LDS image [row][32 k] with the 16-byte column XOR-swizzled by (row>>1)&7, which makes the
//     ds_read_b128 fragment reads of 32 different rows conflict-free (MI355X_MICROARCH.md §LDS).
//   * each lane fetches 4 consecutive k of its row with ONE ds_read_b128 and feeds 4 MFMAs; the
//     k-order inside a chunk is therefore permuted identically for A and B, which a dot product
//     does not care about.  A wave tile of 64x64 needs 4 ds_read_b128 per 16 MFMAs.
//   * zero padding, ragged M and ragged K are resolved in the loader (masked loads of 0).
//   * epilogue: bias -> activation -> (+ residual, optionally through a 2x nearest up-sampling)
//     -> store, plus an optional second output  y*s2 + t2  (the next block's pre-conv BN).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "plan.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == (int)Act::RELU) return v > 0.f ? v : 0.f;
    if (act == (int)Act::PRELU) return v >= 0.f ? v : v * slope;
    if (act == (int)Act::SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(const ConvArgs p, const int tiles_n,
                                                                     const int chunks_total,
                                                                     const int chunks_per_split) {
    constexpr int T = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RP = T / 8;                    // tile rows filled per loader pass
    constexpr int AL = BM / RP, BL = BN / RP;
    static_assert(TM >= 1 && TN >= 1 && AL >= 1 && BL >= 1 && RP % 16 == 0, "tile shape");
    __shared__ v4f lds[2][(BM + BN) * 8];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HoWo = p.Ho * p.Wo;
    const int M = p.B * HoWo;
    const int Ktot = p.ks * p.ks * p.Cin;
    const int c0 = blockIdx.y * chunks_per_split;
    const int c1 = min(chunks_total, c0 + chunks_per_split);

    // ---- loader bookkeeping: this thread fills 16-byte column `lq` of rows lrow + i*RP
    const int lrow = tid >> 3, lq = tid & 7;
    const int sw = lq ^ ((lrow >> 1) & 7);
    int a_base[AL];
    unsigned a_mask[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + lrow + i * RP;
        a_base[i] = 0; a_mask[i] = 0;
        if (m < M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_base[i] = ((n * p.H + iy0) * p.W + ix0) * p.Cin;
            unsigned mk = 0;
            for (int ky = 0; ky < p.ks; ++ky)
                for (int kx = 0; kx < p.ks; ++kx)
                    if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W)
                        mk |= 1u << (ky * p.ks + kx);
            a_mask[i] = mk;
        }
    }
    const float* wrow[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) wrow[i] = p.wt + (size_t)(n0 + lrow + i * RP) * p.Kpad + lq * 4;
    const bool cin32 = (p.Cin & 31) == 0;

    v4f ra[AL], rb[BL];
    auto load_chunk = [&](int kc) {
        const int kb = kc * 32;
        int tap, ci;
        if (cin32) {                              // whole chunk inside one tap (wave-uniform)
            tap = kb / p.Cin;
            ci = kb - tap * p.Cin + lq * 4;
        } else {
            const int k4 = kb + lq * 4;
            tap = k4 / p.Cin;
            ci = k4 - tap * p.Cin;
        }
        const bool kvalid = kb + lq * 4 < Ktot;
        const int ky = tap / p.ks, kx = tap - ky * p.ks;
        const int toff = (ky * p.W + kx) * p.Cin + ci;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (kvalid && ((a_mask[i] >> tap) & 1u)) v = *reinterpret_cast<const v4f*>(p.in + (long)(a_base[i] + toff));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = *reinterpret_cast<const v4f*>(wrow[i] + kb);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AL; ++i) lds[buf][(lrow + i * RP) * 8 + sw] = ra[i];
#pragma unroll
        for (int i = 0; i < BL; ++i) lds[buf][BM * 8 + (lrow + i * RP) * 8 + sw] = rb[i];
    };

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh2 = lane >> 5;
    const int fsw = (fr >> 1) & 7;
    auto compute = [&](int buf) {
        const v4f* A = lds[buf] + (wm * TM * 32 + fr) * 8;
        const v4f* Bt = lds[buf] + BM * 8 + (wn * TN * 32 + fr) * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = (2 * s + fh2) ^ fsw;
            v4f a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = A[i * 32 * 8 + col];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bt[j * 32 * 8 + col];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    };

    if (c0 < c1) {
        load_chunk(c0);
        store_chunk(0);
        __syncthreads();
        for (int kc = c0, it = 0; kc < c1; ++kc, ++it) {
            const int buf = it & 1;
            const bool more = kc + 1 < c1;
            if (more) load_chunk(kc + 1);
            compute(buf);
            if (more) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: column = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + (wn * TN + j) * 32 + fr;
        if (co >= p.Cout) continue;
        if (p.nsplit > 1) {
            float* slab = p.partial + (size_t)blockIdx.y * M * p.Cout;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh2;
                    if (m < M) slab[(size_t)m * p.Cout + co] = acc[i][j][e];
                }
            continue;
        }
        const float bias = p.bias ? p.bias[co] : 0.f;
        const float slope = p.slope ? p.slope[co] : 0.f;
        const float s2 = p.out2 ? p.s2[co] : 0.f, t2 = p.out2 ? p.t2[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh2;
                if (m >= M) continue;
                float v = apply_act(acc[i][j][e] + bias, p.act, slope);
                if (p.res_mode == (int)ResMode::SAME) {
                    v += p.res[(size_t)m * p.Cout + co];
                } else if (p.res_mode == (int)ResMode::UP2X) {
                    const int n = m / HoWo, rem = m - n * HoWo;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    v += p.res[((size_t)(n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout + co];
                }
                if (p.out1) p.out1[(size_t)m * p.Cout + co] = v;
                if (p.out2) p.out2[(size_t)m * p.Cout + co] = v * s2 + t2;
            }
    }
}

// Sums the split-K slabs and applies the same epilogue (used by the 25088-deep FC and by deep,
// small-M convolutions at low batch).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const ConvArgs p) {
    const long M = (long)p.B * p.Ho * p.Wo;
    const long total = M * p.Cout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % p.Cout);
        const long m = idx / p.Cout;
        float acc = 0.f;
        for (int s = 0; s < p.nsplit; ++s) acc += p.partial[(size_t)s * total + idx];
        float v = apply_act(acc + (p.bias ? p.bias[co] : 0.f), p.act, p.slope ? p.slope[co] : 0.f);
        if (p.res_mode == (int)ResMode::SAME) {
            v += p.res[idx];
        } else if (p.res_mode == (int)ResMode::UP2X) {
            const int HoWo = p.Ho * p.Wo;
            const int n = (int)(m / HoWo), rem = (int)(m - (long)n * HoWo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            v += p.res[((size_t)(n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout + co];
        }
        if (p.out1) p.out1[idx] = v;
        if (p.out2) p.out2[idx] = v * p.s2[co] + p.t2[co];
    }
}

int conv_wt_rows(int Cout) { return (Cout + 127) / 128 * 128; }

int conv_pick_cfg(long M, int Cout) {
    // padded-column waste per candidate tile width; prefer the wider tile when it costs <= 10 % more
    auto cols = [&](int bn) { return (Cout + bn - 1) / bn * bn; };
    int bn = 32;
    if (cols(64) * 10 <= cols(32) * 11) bn = 64;
    if (cols(128) * 10 <= cols(bn) * 11) bn = 128;
    if (bn == 128) return 0;
    if (bn == 64) return (M / 256) * (cols(64) / 64) >= 1024 ? 1 : 3;
    return 2;
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const ConvArgs& a, hipStream_t s) {
    const long M = (long)a.B * a.Ho * a.Wo;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (a.Cout + BN - 1) / BN;
    const int chunks = a.Kpad / 32;
    const int nsplit = a.nsplit < 1 ? 1 : a.nsplit;
    const int cps = (chunks + nsplit - 1) / nsplit;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nsplit);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), 0, s, a, tiles_n, chunks, cps);
}

void launch_conv(const ConvArgs& a, int cfg, hipStream_t s) {
    const long M = (long)a.B * a.Ho * a.Wo;
    if (M <= 0) return;
    if (cfg < 0) cfg = conv_pick_cfg(M, a.Cout);
    switch (cfg) {
        case 0: launch_cfg<128, 128, 2, 2>(a, s); break;
        case 1: launch_cfg<256, 64, 4, 1>(a, s); break;
        case 2: launch_cfg<128, 32, 4, 1>(a, s); break;
        default: launch_cfg<64, 64, 2, 2>(a, s); break;
    }
    if (a.nsplit > 1) launch_splitk_finish(a, s);
}

void launch_splitk_finish(const ConvArgs& a, hipStream_t s) {
    const long total = (long)a.B * a.Ho * a.Wo * a.Cout;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace fh
