// conv_wino2.hip — 3x3 stride-1 pad-1 convolutions with few input channels (Cin = 32 .. 96: IResNet's stage 1, where an unfused
// Winograd pass does not pay) as ONE fused Winograd F(2x2, 3x3) kernel on v_mfma_f32_32x32x2_f32 (gfx950 / CDNA4).
//
// What it replaces: the Conv nodes of w600k_r50's 64-channel stages inside `session_->Run` (reference src/face_recognizer.cpp:279-283)
// — 112x112x64 -> 64, 4 x 56x56x64 -> 64, 56x56x64 -> 128: 2.7 ms of the 12.1 ms step at B = 128 in the direct form (conv_tall_kernel,
// 113-115 TFLOP/s = 0.73 of the f32 MFMA peak: at its ceiling).  Y = A^T [ (G g G^T) (.) (B^T d B) ] A on 2x2 output tiles needs 16
// multiplies per 4 outputs and input channel instead of 36: 2.25x less matrix-core work.  The unfused form (transform kernel -> GEMM ->
// transform kernel, as winograd.hip does for Cin >= 128) moves 4x the activation through memory twice around a K = 64 GEMM that is all
// prologue; here nothing but the layer's input and output touches memory:
//
//   * a wave owns a TILE GROUP of 4 x TGC (7 or 8) output tiles = one tile per MFMA column, and 32 of the 64 output channels of the
//     workgroup's column tile; a workgroup = 2 tile groups x 2 channel halves (4 waves), two workgroups per CU;
//   * the (2*4+2) x (2*TGC+2) input halo of a tile group comes in by LDS-DMA, one 32-channel chunk at a time, de-interleaved into the
//     four (row parity, column parity) planes — tile (tr, tc) reads patch pixel (dy, dx) from plane (dy & 1, dx & 1) at row
//     (tr + dy/2) * PW + tc + dx/2: lanes of one ds_read_b128 touch consecutive rows of a [row][8 x float4] image whose 16-byte column
//     is XOR-swizzled per row, conflict-free (scripts/wino2_banks.py enumerates every access of both configurations);
//   * for each of the 16 frequencies f = (i, j) in turn: the lane forms V_f = (B^T d B)[i][j] of ITS tile for 4 channels from four
//     ds_read_b128 and three packed adds (B^T has two +-1 entries per row) — a float4 that is exactly the B fragment the MFMA wants
//     from that lane; U_f (64 x 32 per chunk, 8 KB, pre-arranged in fragment order) streams through a double-buffered LDS stage;
//     16 MFMAs accumulate M_f, and M_f is added into the four output accumulators Y[a][b] with A^T's {0, +-1} coefficients
//     (1, 2 or 4 adds per register) while the NEXT frequency's MFMAs run;
//   * epilogue on the 2x2 pixels of the lane's tile: bias (9 border classes when the block's BatchNorm is folded in, engine.cpp) ->
//     ReLU / PReLU -> + residual -> store (+ second output), per-channel vectors parked in LDS, residual loads before the stores.
//
// Numerics: the interpolation points of F(2x2, 3x3) are {0, 1, -1, inf}; G has entries 1 and 1/2 — rounding stays within ~3x of the
// direct fp32 form (tests/test_gpu_round4.py bars single layers at 5e-5 abs on O(1) outputs; winograd.hip's F(4x4) needs 2e-4).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "kernels.h"
#include "plan.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

// global -> LDS without a register round trip: the global address is per lane, the LDS address is the WAVE-UNIFORM dst + 16 * lane
__device__ __forceinline__ void dma16(const float* src, void* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

// NONE / ReLU / PReLU without a branch per element: sl = 1 (none) or the PReLU slope; relu is wave-uniform
__device__ __forceinline__ float act1(float v, bool relu, float sl) {
    const float a = v >= 0.f ? v : v * sl;
    return relu ? fmaxf(v, 0.f) : a;
}

// B^T of F(2x2, 3x3): row i has two non-zero entries, at patch rows YA[i] (sign SA[i]) and YB[i] (sign SB[i])
//   [ 1  0 -1  0 ]   [ 0  1  1  0 ]   [ 0 -1  1  0 ]   [ 0  1  0 -1 ]
constexpr int kYA[4] = {0, 1, 2, 1}, kYB[4] = {2, 2, 1, 3};
constexpr int kSA[4] = {1, 1, 1, 1}, kSB[4] = {-1, 1, -1, -1};
// A^T = [ 1 1 1 0 ; 0 1 -1 -1 ]: coefficient of frequency row i in output row a
constexpr int kAT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};

template <int TGC> struct Geo {
    static constexpr int PW = TGC + 1;                 // plane width (pixels of one column parity)
    static constexpr int PH = 5;                       // plane height: 4 tile rows + 1
    static constexpr int RPT = 4 * PH * PW;            // LDS rows (= pixels) of one tile group's halo: 160 / 180
    static constexpr int NP = (2 * RPT + 31) / 32;     // loader passes of a workgroup (32 rows x 8 float4 per pass): 10 / 12
    static constexpr int HALO_BYTES = NP * 32 * 128;
};

// swizzle key of LDS row (plane pl, plane row pr, plane column pc): the 16-byte column c of the pixel lives at column c ^ key
template <int TGC> __device__ __forceinline__ int row_key(int pl, int pr, int pc) {
    if (TGC == 7) return ((pc >> 1) + 4 * (pr + pl)) & 7;            // = (row >> 1) & 7 with PW = 8: rows of a read are consecutive mod 16
    return ((pc >> 1) + 2 * pr) & 7;                                  // PW = 9: tile rows {0, 2} / {1, 3} share a 16-lane read group
}

// lane (0..31) -> tile of the group.  TGC = 7: 8 lanes per tile row, the eighth idles.  TGC = 8: the two 16-lane groups a ds_read_b128
// is served in ({0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}) take tile rows {0, 2} and {1, 3}, whose keys differ by 4.
template <int TGC> __device__ __forceinline__ void lane_tile(int t, int& tr, int& tc, bool& live) {
    if (TGC == 7) { tr = t >> 3; tc = t & 7; live = tc < 7; if (!live) tc = 6; return; }
    live = true;
    if (t < 4) { tr = 0; tc = t; }
    else if (t < 12) { tr = 1; tc = t - 4; }
    else if (t < 16) { tr = 0; tc = t - 8; }
    else if (t < 20) { tr = 3; tc = t - 16; }
    else if (t < 28) { tr = 2; tc = t - 20; }
    else { tr = 3; tc = t - 24; }
}

// One workgroup: tile groups 2*pair, 2*pair + 1 (linear over batch x group rows x group columns) x output channels [64 tile_n, +64).
template <int TGC>
__global__ __launch_bounds__(256, 2) void wino2_kernel(const ConvArgs p, const int tiles_n, const int tgx, const int tgy, const int n_tg) {
    using G = Geo<TGC>;
    constexpr int PW = G::PW, PH = G::PH, RPT = G::RPT, NP = G::NP;
    extern __shared__ v4f w2sm[];
    char* const halo = reinterpret_cast<char*>(w2sm);                      // [2 groups][RPT rows][8 x 16 B]
    char* const wbuf = halo + G::HALO_BYTES;                               // [2 buffers][8 KB]: U_f of one 32-channel chunk, fragment order

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tgi = wid >> 1, mb = wid & 1;                                // this wave's tile group / 32-channel half of the column tile
    int blk;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, x = blockIdx.x & 7;
        blk = x * q + min(x, r) + (int)(blockIdx.x >> 3);                  // XCD-contiguous order: neighbouring groups share an L2
    }
    const int tile_n = blk % tiles_n, pair = blk / tiles_n;
    const int n0 = tile_n * 64;
    const int H = p.H, W = p.W, Cin = p.Cin;
    const int per_img = tgx * tgy;

    // ---- halo loader: pass i fills LDS rows i * 32 + (tid >> 3), physical column tid & 7
    unsigned a_off[NP];                                                    // element offset into p.in, ~0u = the zero line
    {
        const int col = tid & 7;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int R = i * 32 + (tid >> 3);
            const int g = R / RPT, rr = R - g * RPT;
            const int pl = rr / (PH * PW), r2 = rr - pl * (PH * PW);
            const int pr = r2 / PW, pc = r2 - pr * PW;
            const int tg = 2 * pair + g;
            const int n = tg / per_img, rem = tg - n * per_img;
            const int gy = rem / tgx, gx = rem - gy * tgx;
            const int y = 8 * gy - 1 + 2 * pr + (pl >> 1), x = 2 * TGC * gx - 1 + 2 * pc + (pl & 1);
            const bool ok = g < 2 && tg < n_tg && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int lc = col ^ row_key<TGC>(pl, pr, pc);
            a_off[i] = ok ? (unsigned)((((size_t)n * H + y) * W + x) * Cin + lc * 4) : ~0u;
        }
    }
    // ---- weight stage loader: stage s = chunk * 16 + f is 8 KB = 8 wave pieces of 1 KB; wave w moves pieces 2w, 2w + 1
    const float* w_src = p.wt + (size_t)tile_n * (Cin >> 5) * 16 * 2048 + (wid * 2) * 256 + lane * 4;
    char* const w_dst = wbuf + wid * 2048;
    auto load_w = [&](int s, int buf) __attribute__((always_inline)) {
        const float* src = w_src + (size_t)s * 2048;
        char* dst = w_dst + buf * 8192;
        dma16(src, dst);
        dma16(src + 256, dst + 1024);
    };

    // ---- this lane's tile and the LDS byte offsets of its 16 patch pixels (column bits: key ^ half; the k-group g adds ^ 32 g)
    const int t = lane & 31, h = lane >> 5;
    int tr, tc; bool live;
    lane_tile<TGC>(t, tr, tc, live);
    unsigned rb[16];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int pl = (dy & 1) * 2 + (dx & 1), pr = tr + (dy >> 1), pc = tc + (dx >> 1);
            const int R = tgi * RPT + (pl * PH + pr) * PW + pc;
            rb[dy * 4 + dx] = (unsigned)(R * 128 + ((row_key<TGC>(pl, pr, pc) ^ h) << 4));
        }
    const char* const wfrag = wbuf + (mb * 4 * 64 + lane) * 16;            // + buffer * 8192 + g * 1024

    v16f Y[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[a][b][e] = 0.f;

    // the epilogue's per-channel vectors ([9 bias classes | slope | s2 | t2] x 64), fetched now, parked in LDS after the K loop
    float epv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = tid + k * 256, a = e >> 6, co = n0 + (e & 63);
        float v = 0.f;
        if (co < p.Cout) {
            if (a < 9) { if (p.bias && (a == 0 || p.bias_cls)) v = p.bias[a * p.Cout + co]; }
            else if (a == 9) v = p.act == (int)Act::PRELU ? p.slope[co] : 1.f;
            else if (p.out2) v = a == 10 ? p.s2[co] : p.t2[co];
        }
        epv[k] = v;
    }

    // Y[a][b] += AT[a][i] * AT[b][j] * M   (frequency f = 4 i + j)
    auto y_update = [&](const int f, const v16f& M) __attribute__((always_inline)) {
        const int i = f >> 2, j = f & 3;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int cf = kAT[a][i] * kAT[b][j];
                if (cf > 0) Y[a][b] += M;
                else if (cf < 0) Y[a][b] -= M;
            }
    };

    const int NC = Cin >> 5, NS = NC * 16;
    v16f Mprev;
#pragma unroll
    for (int e = 0; e < 16; ++e) Mprev[e] = 0.f;
    load_w(0, 0);
    for (int c = 0; c < NC; ++c) {
        if (c > 0) __syncthreads();                                        // every wave is done with the previous chunk's halo
#pragma unroll
        for (int i = 0; i < NP; ++i)
            dma16(a_off[i] != ~0u ? p.in + a_off[i] + c * 32 : p.zeros, halo + (i * 32 + wid * 8) * 128);
        __syncthreads();                                                   // (drains vmcnt: halo chunk + this stage's weights have landed)
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            // next stage's weights into the other buffer (the very last stage re-fetches itself: no branch in the loop body)
            load_w(min(c * 16 + f + 1, NS - 1), (f + 1) & 1);
            const int fi = f >> 2, fj = f & 3;
            const int ya = kYA[fi], yb = kYB[fi], xa = kYA[fj], xb = kYB[fj];
            const int saa = kSA[fi] * kSA[fj], sab = kSA[fi] * kSB[fj], sba = kSB[fi] * kSA[fj], sbb = kSB[fi] * kSB[fj];
            const char* const wf = wfrag + (f & 1) * 8192;
            const unsigned r00 = rb[ya * 4 + xa], r01 = rb[ya * 4 + xb], r10 = rb[yb * 4 + xa], r11 = rb[yb * 4 + xb];
            v16f M;
#pragma unroll
            for (int e = 0; e < 16; ++e) M[e] = 0.f;
            v4f d[2][4], w[2];
            auto fetch = [&](int g) __attribute__((always_inline)) {
                d[g & 1][0] = *reinterpret_cast<const v4f*>(halo + (r00 ^ (g << 5)));
                d[g & 1][1] = *reinterpret_cast<const v4f*>(halo + (r01 ^ (g << 5)));
                d[g & 1][2] = *reinterpret_cast<const v4f*>(halo + (r10 ^ (g << 5)));
                d[g & 1][3] = *reinterpret_cast<const v4f*>(halo + (r11 ^ (g << 5)));
                w[g & 1] = *reinterpret_cast<const v4f*>(wf + g * 1024);
            };
            fetch(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) fetch(g + 1);                                   // the next k-group's fragments are in flight behind this one's MFMAs
                const v4f* dd = d[g & 1];
                v4f v = saa > 0 ? dd[0] : -dd[0];
                v = sab > 0 ? v + dd[1] : v - dd[1];
                v = sba > 0 ? v + dd[2] : v - dd[2];
                v = sbb > 0 ? v + dd[3] : v - dd[3];
#pragma unroll
                for (int e = 0; e < 4; ++e) M = __builtin_amdgcn_mfma_f32_32x32x2f32(w[g & 1][e], v[e], M, 0, 0, 0);
            }
            // the previous frequency's result goes into the outputs while this one's MFMAs run (Mprev = 0 in front of the first stage)
            y_update((f + 15) & 15, Mprev);
            Mprev = M;
            if (f < 15) __syncthreads();                                   // next weight stage landed; everyone is done with this one
        }
    }
    y_update(15, Mprev);
    __syncthreads();                                                       // K loop over: the weight buffers are free
    float* const ep = reinterpret_cast<float*>(wbuf);
#pragma unroll
    for (int k = 0; k < 3; ++k) ep[tid + k * 256] = epv[k];
    __syncthreads();

    // ---- epilogue: lane = tile (tr, tc) of group 2 pair + tgi, accumulator quad q = channels n0 + 32 mb + 8 q + 4 h .. + 3
    const int tg = 2 * pair + tgi;
    if (!live || tg >= n_tg) return;
    const int n = tg / per_img, rem = tg - n * per_img;
    const int gy = rem / tgx, gx = rem - gy * tgx;
    const int oy0 = 2 * (4 * gy + tr), ox0 = 2 * (TGC * gx + tc);
    const int cl0 = 32 * mb + 4 * h;                                       // channel within the column tile (quad 0)
    const float* __restrict__ res = p.res;
    float* __restrict__ out1 = p.out1;
    float* __restrict__ out2 = p.out2;
    const bool relu = p.act == (int)Act::RELU;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oy = oy0 + a, ox = ox0 + b;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            const size_t row = (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.Cout + n0;
            const int cls = p.bias_cls ? 3 * (oy == 0 ? 0 : oy == p.Ho - 1 ? 2 : 1) + (ox == 0 ? 0 : ox == p.Wo - 1 ? 2 : 1) : 0;
            v4f r4[4];
            if (p.res_mode != (int)ResMode::NONE) {
#pragma unroll
                for (int q = 0; q < 4; ++q) r4[q] = *reinterpret_cast<const v4f*>(res + row + cl0 + 8 * q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = cl0 + 8 * q;
                const v4f b4 = *reinterpret_cast<const v4f*>(ep + cls * 64 + cl);
                const v4f sl = *reinterpret_cast<const v4f*>(ep + 9 * 64 + cl);
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act1(Y[a][b][4 * q + e] + b4[e], relu, sl[e]);
                if (p.res_mode != (int)ResMode::NONE) v += r4[q];
                if (out1) *reinterpret_cast<v4f*>(out1 + row + cl) = v;
                if (out2) {
                    const v4f s2 = *reinterpret_cast<const v4f*>(ep + 10 * 64 + cl), t2 = *reinterpret_cast<const v4f*>(ep + 11 * 64 + cl);
                    *reinterpret_cast<v4f*>(out2 + row + cl) = v * s2 + t2;
                }
            }
        }
}

int pick_tgc(int W) {
    const int wt = (W + 1) / 2;                                            // tile columns of the map
    const int g7 = (wt + 6) / 7 * 8, g8 = (wt + 7) / 8 * 8;                // MFMA columns spent per tile row (a 7-wide group idles its eighth lane)
    return g7 < g8 ? 7 : 8;
}

template <int TGC>
void launch_tgc(const ConvArgs& a, hipStream_t s) {
    const int ht = (a.H + 1) / 2, wt = (a.W + 1) / 2;
    const int tgy = (ht + 3) / 4, tgx = (wt + TGC - 1) / TGC;
    const long n_tg = (long)a.B * tgy * tgx;
    const int tiles_n = a.Cout / 64;
    const long blocks = (n_tg + 1) / 2 * tiles_n;
    const size_t lds = Geo<TGC>::HALO_BYTES + 2 * 8192;
    static bool attr_done = false;
    if (!attr_done) {
        FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino2_kernel<TGC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL((wino2_kernel<TGC>), dim3((unsigned)blocks), dim3(256), lds, s, a, tiles_n, tgx, tgy, (int)n_tg);
    // booked with the FLOPs the matrix cores EXECUTE (16 products per 2x2 tile and channel pair, idle lanes included)
    timer.end(s, 12, 2.0 * 16 * 32.0 * (double)((n_tg + 1) / 2 * 2) * a.Cin * a.Cout, a.t_flops);   // (bytes slot: the layer's direct-form FLOPs, as tag 7)
}

}  // namespace

// U_f = G g G^T per (output channel, input channel) in fp64, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], laid out in the order the kernel's
// LDS stages and MFMA fragments want: [Cout / 64][Cin / 32][16 f][2 halves of 32 channels][4 k-groups][2 k-halves][32 rows][4 floats],
// element = U_f[cout = 64 tn + 32 mb + m][cin = 32 c + 8 g + 4 kh + e].  w = [Cout][9 taps][Cin] (the engine's layout).
size_t wino2_weight_floats(int Cin, int Cout) { return (size_t)16 * Cin * Cout; }
void wino2_pack_weights(const float* w, int Cout, int Cin, float* dst) {
    static const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    const int NC = Cin / 32;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            double g[3][3], t[4][3];
            for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = w[((size_t)co * 9 + k) * Cin + ci];
            for (int i = 0; i < 4; ++i)
                for (int x = 0; x < 3; ++x) t[i][x] = Gm[i][0] * g[0][x] + Gm[i][1] * g[1][x] + Gm[i][2] * g[2][x];
            const int tn = co / 64, mbh = (co % 64) / 32, m = co % 32;
            const int c = ci / 32, gq = (ci % 32) / 8, kh = (ci % 8) / 4, e = ci % 4;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = t[i][0] * Gm[j][0] + t[i][1] * Gm[j][1] + t[i][2] * Gm[j][2];
                    const size_t stage = ((size_t)tn * NC + c) * 16 + (i * 4 + j);
                    dst[stage * 2048 + ((((size_t)mbh * 4 + gq) * 2 + kh) * 32 + m) * 4 + e] = (float)u;
                }
        }
}

bool wino2_ok(const ConvArgs& a) {
    return a.ks == 3 && a.stride == 1 && a.pad == 1 && a.H == a.Ho && a.W == a.Wo && a.H >= 2 && a.W >= 2 && a.Cin % 32 == 0 && a.Cin >= 32 &&
           a.Cout % 64 == 0 && a.act != (int)Act::SIGMOID && a.n_outs == 0 && !a.sc_in && !a.dw_w && !a.u8_src && a.wt_group_rows == 0 &&
           (a.res_mode == (int)ResMode::NONE || a.res_mode == (int)ResMode::SAME);
}

long wino2_blocks(const ConvArgs& a) {
    const int tgc = pick_tgc(a.W);
    const long n_tg = (long)a.B * (((a.H + 1) / 2 + 3) / 4) * (((a.W + 1) / 2 + tgc - 1) / tgc);
    return (n_tg + 1) / 2 * (a.Cout / 64);
}

// a.wt = wino2_pack_weights' image of the filter; everything else as for launch_conv
void launch_wino2(const ConvArgs& a_in, hipStream_t s) {
    if (!wino2_ok(a_in)) throw std::runtime_error("launch_wino2: layer shape not supported");
    ConvArgs a = a_in;
    a.zeros = conv_zero_line();
    if (pick_tgc(a.W) == 7) launch_tgc<7>(a, s);
    else launch_tgc<8>(a, s);
}

}  // namespace fh
