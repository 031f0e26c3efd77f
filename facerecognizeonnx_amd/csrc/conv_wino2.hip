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
//     from that lane; U_f (pre-arranged in fragment order) goes global -> registers, one frequency ahead (no barrier inside a chunk);
//     16 MFMAs accumulate M_f, and M_f is added into the four output accumulators Y[a][b] with A^T's {0, +-1} coefficients
//     (1, 2 or 4 adds per register) while the NEXT frequency's MFMAs run;
//   * epilogue on the 2x2 pixels of the lane's tile: bias (9 border classes when the block's BatchNorm is folded in, engine.cpp) ->
//     ReLU / PReLU -> + residual -> store (+ second output), per-channel vectors parked in LDS, residual loads before the stores.
//
// Numerics: the interpolation points of F(2x2, 3x3) are {0, 1, -1, inf}; G has entries 1 and 1/2 — rounding stays within ~3x of the
// direct fp32 form (tests/test_gpu_round4.py bars single layers at 5e-5 abs on O(1) outputs; winograd.hip's F(4x4) needs 2e-4).
#include <hip/hip_runtime.h>

#include <algorithm>
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

template <int V> struct IC { static constexpr int value = V; };          // compile-time integers for the fold over frequencies

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
// The weight fragments go global -> registers (each wave fetches the 4 KB of U_f it multiplies, one frequency ahead; the two waves of a
// workgroup that share a channel half hit the same lines in L1), not through an LDS stage shared by the workgroup (measured: 204 vs 197 us
// on 56x56x64 at B = 128): no barrier inside a 32-channel chunk, the waves of a workgroup run apart and overlap each other's phases.
template <int TGC>
__global__ __launch_bounds__(256, 2) void wino2_kernel(const ConvArgs p, const int tiles_n, const int tgx, const int tgy, const int n_tg) {
    using G = Geo<TGC>;
    constexpr int PW = G::PW, PH = G::PH, RPT = G::RPT, NP = G::NP;
    extern __shared__ v4f w2sm[];
    char* const halo = reinterpret_cast<char*>(w2sm);                      // [2 groups][RPT rows][8 x 16 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tgi = wid >> 1, mb = wid & 1;                                // this wave's tile group / 32-channel half of the column tile
#ifdef FACEHIP_W2_PROF
    unsigned w2st[8] = {0, 0, 0, 0, 0, 0, 0, 0};                           // (32-bit, constant indices only: eight scalar registers)
    const long long w2rt0 = __builtin_amdgcn_s_memrealtime();
#define W2_STAMP(i) { w2st[i] = (unsigned)__builtin_readcyclecounter(); }
#else
#define W2_STAMP(i)
#endif
    W2_STAMP(0)
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
    // (image, group row, group column) of the workgroup's two tile groups: wave-uniform, the second is the first's successor — the two
    // runtime divisions happen once per workgroup on scalars, not per lane and loader pass
    int gn[2], ggy[2], ggx[2];
    {
        const int tg0 = 2 * pair;
        const int n = tg0 / per_img, rem = tg0 - n * per_img;
        const int gy = rem / tgx;
        gn[0] = __builtin_amdgcn_readfirstlane(n); ggy[0] = __builtin_amdgcn_readfirstlane(gy); ggx[0] = __builtin_amdgcn_readfirstlane(rem - gy * tgx);
        gn[1] = gn[0]; ggy[1] = ggy[0]; ggx[1] = ggx[0] + 1;
        if (ggx[1] == tgx) { ggx[1] = 0; if (++ggy[1] == tgy) { ggy[1] = 0; ++gn[1]; } }
    }
    {
        const int col = tid & 7;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int R = i * 32 + (tid >> 3);
            const int g = R / RPT, rr = R - g * RPT;
            const int pl = rr / (PH * PW), r2 = rr - pl * (PH * PW);
            const int pr = r2 / PW, pc = r2 - pr * PW;
            const int n = g ? gn[1] : gn[0], gy = g ? ggy[1] : ggy[0], gx = g ? ggx[1] : ggx[0];
            const int y = 8 * gy - 1 + 2 * pr + (pl >> 1), x = 2 * TGC * gx - 1 + 2 * pc + (pl & 1);
            const bool ok = g < 2 && 2 * pair + g < n_tg && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            const int lc = col ^ row_key<TGC>(pl, pr, pc);
            a_off[i] = ok ? (unsigned)((((size_t)n * H + y) * W + x) * Cin + lc * 4) : ~0u;
        }
    }
    // ---- this lane's tile and the LDS byte offsets of its 16 patch pixels (column bits: key ^ half; the k-group g adds ^ 32 g)
    const int t = lane & 31, h = lane >> 5;
    int tr, tc; bool live;
    lane_tile<TGC>(t, tr, tc, live);
    typedef const __attribute__((address_space(3))) v4f* lds_v4f;
    const unsigned halo_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)halo;
    unsigned rb[16];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int pl = (dy & 1) * 2 + (dx & 1), pr = tr + (dy >> 1), pc = tc + (dx >> 1);
            const int R = tgi * RPT + (pl * PH + pr) * PW + pc;
            rb[dy * 4 + dx] = halo_base + (unsigned)(R * 128 + ((row_key<TGC>(pl, pr, pc) ^ h) << 4));   // (halo_base % 128 == 0: XOR-safe)
        }

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

    // Y[a][b] += AT[a][i] * AT[b][j] * M   (frequency f = 4 i + j; whole accumulators: updating them register by register behind each
    // MFMA was tried — the compiler then keeps eight stages' M tuples alive and sums them late: 300 spilled registers, 3x slower)
    auto y_update = [&](auto fc, const v16f& M) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value, i = f >> 2, j = f & 3;
        constexpr int c00 = kAT[0][i] * kAT[0][j], c01 = kAT[0][i] * kAT[1][j], c10 = kAT[1][i] * kAT[0][j], c11 = kAT[1][i] * kAT[1][j];
        if constexpr (c00 > 0) Y[0][0] += M; else if constexpr (c00 < 0) Y[0][0] -= M;
        if constexpr (c01 > 0) Y[0][1] += M; else if constexpr (c01 < 0) Y[0][1] -= M;
        if constexpr (c10 > 0) Y[1][0] += M; else if constexpr (c10 < 0) Y[1][0] -= M;
        if constexpr (c11 > 0) Y[1][1] += M; else if constexpr (c11 < 0) Y[1][1] -= M;
        // (pinned: a deferred update keeps its frequency's 16 accumulator registers alive — see wino2x_kernel)
        if constexpr (c00 != 0) asm volatile("" : "+v"(Y[0][0]));
        if constexpr (c01 != 0) asm volatile("" : "+v"(Y[0][1]));
        if constexpr (c10 != 0) asm volatile("" : "+v"(Y[1][0]));
        if constexpr (c11 != 0) asm volatile("" : "+v"(Y[1][1]));
    };
    const int NC = Cin >> 5, NS = NC * 16;
#ifdef FACEHIP_W2_PROF
    const bool abl_no_halo = p.sk_test_drop & 1;                          // ablations (FACEHIP_W2_ABLATE bits): 1 = skip the halo DMA (stale LDS),
    const bool abl_no_store = p.sk_test_drop & 2;                         // 2 = skip the epilogue's loads and stores
#else
    constexpr bool abl_no_halo = false;
#endif
    v16f Mprev;
#pragma unroll
    for (int e = 0; e < 16; ++e) Mprev[e] = 0.f;
    // One frequency f (a compile-time constant: the loop over f is a fold over 16 instantiations — `#pragma unroll` gives up on a body
    // of this size and a run-time f turns every B^T / A^T coefficient into a branch): 4 k-groups x 4 MFMAs, the patch pixels of the next
    // step fetched one step ahead, the previous frequency's accumulator register m folded into the outputs right behind MFMA m.
    v4f d[2][4];
    auto fetch_d = [&](auto fc, auto gc, auto bc) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value, g = decltype(gc)::value, buf = decltype(bc)::value;
        constexpr int fi = f >> 2, fj = f & 3;
        constexpr int ya = kYA[fi], yb = kYB[fi], xa = kYA[fj], xb = kYB[fj];
        // rb[] are absolute LDS addresses (no base add per read); the k-group's column bits come from a scalar the optimiser cannot see
        // through — otherwise the 16 x 3 XORed addresses are computed once and kept in 48 registers for the whole loop
        unsigned gx = g << 5;
        asm volatile("" : "+s"(gx));                                       // (for g = 0 too: reads that depend on nothing are hoisted stages ahead)
        d[buf][0] = *reinterpret_cast<lds_v4f>((size_t)(rb[ya * 4 + xa] ^ gx));
        d[buf][1] = *reinterpret_cast<lds_v4f>((size_t)(rb[ya * 4 + xb] ^ gx));
        d[buf][2] = *reinterpret_cast<lds_v4f>((size_t)(rb[yb * 4 + xa] ^ gx));
        d[buf][3] = *reinterpret_cast<lds_v4f>((size_t)(rb[yb * 4 + xb] ^ gx));
    };
    v4f wr[2][4];                                                          // weight fragments: [stage parity][k-group]
    const float* const wg_src = p.wt + (size_t)tile_n * (Cin >> 5) * 16 * 2048 + (mb * 4 * 64 + lane) * 4;   // this wave's 4 KB of stage 0
    auto fetch_w = [&](int s, int buf) __attribute__((always_inline)) {   // global -> registers, 4 x (64 lanes x 16 B), k-group g at + g KB
        const float* src = wg_src + (size_t)s * 2048;
#pragma unroll
        for (int g = 0; g < 4; ++g) wr[buf][g] = *reinterpret_cast<const v4f*>(src + g * 256);
    };
    int c = 0;
    auto stage = [&](auto fc) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value;
        constexpr int fi = f >> 2, fj = f & 3;
        constexpr int saa = kSA[fi] * kSA[fj], sab = kSA[fi] * kSB[fj], sba = kSB[fi] * kSA[fj], sbb = kSB[fi] * kSB[fj];
        // the next stage's weights (the very last stage re-fetches itself: no branch in the loop body)
        fetch_w(min(c * 16 + f + 1, NS - 1), (f + 1) & 1);
        v16f M;
#pragma unroll
        for (int e = 0; e < 16; ++e) M[e] = 0.f;
        auto step = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, k = f * 4 + g;
            if constexpr (g < 3) fetch_d(IC<f>{}, IC<g + 1>{}, IC<(k + 1) & 1>{});
            else if constexpr (f < 15) fetch_d(IC<f + 1>{}, IC<0>{}, IC<(k + 1) & 1>{});
            __builtin_amdgcn_sched_barrier(0);                             // (reads issued BEFORE the MFMAs: left alone the scheduler sinks them)
            const v4f* dd = d[k & 1];
            v4f v = saa > 0 ? dd[0] : -dd[0];
            v = sab > 0 ? v + dd[1] : v - dd[1];
            v = sba > 0 ? v + dd[2] : v - dd[2];
            v = sbb > 0 ? v + dd[3] : v - dd[3];
            asm volatile("" : "+v"(v));                                    // (pins all of V in front of the MFMAs: otherwise each MFMA is preceded
            __builtin_amdgcn_sched_barrier(0);                             //  by its three adds and a VALU -> MFMA-operand nop)
#pragma unroll
            for (int e = 0; e < 4; ++e) M = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[f & 1][g][e], v[e], M, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{});
        y_update(IC<(f + 15) & 15>{}, Mprev);                              // the previous frequency's result -> the outputs, while this one's MFMAs run
        Mprev = M;                                                         // (Mprev = 0 in front of the very first stage)
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch_w(0, 0);
    for (c = 0; c < NC; ++c) {
        if (c > 0) __syncthreads();                                        // every wave is done with the previous chunk's halo
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (!abl_no_halo) dma16(a_off[i] != ~0u ? p.in + a_off[i] + c * 32 : p.zeros, halo + (i * 32 + wid * 8) * 128);
        __syncthreads();                                                   // (drains vmcnt: the halo chunk has landed)
        fetch_d(IC<0>{}, IC<0>{}, IC<0>{});
        stage(IC<0>{}); stage(IC<1>{}); stage(IC<2>{}); stage(IC<3>{}); stage(IC<4>{}); stage(IC<5>{}); stage(IC<6>{}); stage(IC<7>{});
        stage(IC<8>{}); stage(IC<9>{}); stage(IC<10>{}); stage(IC<11>{}); stage(IC<12>{}); stage(IC<13>{}); stage(IC<14>{}); stage(IC<15>{});
    }
    y_update(IC<15>{}, Mprev);
    __syncthreads();                                                       // K loop over: the halo image is free
#ifdef FACEHIP_W2_PROF
    auto w2_flush = [&]() {
        W2_STAMP(7)
        if (lane == 0 && p.slabs && blockIdx.x < 4096) {                   // [workgroup][wave][8]: stamps 1..7 relative to stamp 0; [7] = shader MHz x 10
            long long* o = reinterpret_cast<long long*>(p.slabs) + ((size_t)blockIdx.x * 4 + wid) * 8;
#pragma unroll
            for (int i = 1; i < 8; ++i) o[i - 1] = (long long)(unsigned)(w2st[i] - w2st[0]);
            o[7] = (long long)(unsigned)(w2st[7] - w2st[0]) * 1000 / ((long long)__builtin_amdgcn_s_memrealtime() - w2rt0 + 1);
        }
    };
#endif
    float* const ep = reinterpret_cast<float*>(halo);
#pragma unroll
    for (int k = 0; k < 3; ++k) ep[tid + k * 256] = epv[k];
    __syncthreads();

    // ---- epilogue: lane = tile (tr, tc) of group 2 pair + tgi, accumulator quad q = channels n0 + 32 mb + 8 q + 4 h .. + 3
    const int tg = 2 * pair + tgi;
#ifdef FACEHIP_W2_PROF
    W2_STAMP(6)                                                            // K loop + barriers + epilogue vectors in LDS
    if (!live || tg >= n_tg || abl_no_store) { w2_flush(); return; }
#else
    if (!live || tg >= n_tg) return;
#endif
    const int n = gn[tgi], gy = ggy[tgi], gx = ggx[tgi];
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
#ifdef FACEHIP_W2_PROF
    w2_flush();
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// wino2x_kernel — the same algorithm on v_mfma_f32_16x16x4_f32 for layers with exactly 64 input channels: ONE WAVE per workgroup owns a
// 4 x 4 tile group (8 x 8 outputs = the 16 MFMA columns) and ALL 16 CB output channels of its column tile.
//   * 16 tiles per wave tile every map whose side is a multiple of 8 exactly (56 = 7 x 8: the 32-column form idles an eighth of its lanes
//     there) and let one wave cover 64 channels, so V_f is formed ONCE per tile (the 32-column form computes it in both channel halves):
//     per 1024 cycles of matrix-pipe time 24 + 18 VALU for V and the output update instead of 48 + 36, 8 ds_read_b128 instead of 16;
//   * the whole 64-channel halo of the group (10 x 10 pixels x 256 B = 25.6 KB, four parity planes of 5 x 5 pixels, 16-byte column
//     XOR-swizzled by 2 ((pc + 4 (pr & 1)) & 7): conflict-free, scripts/wino2_banks.py) is loaded once: no chunk switch, and — a wave
//     only reads what it loaded itself — NO barrier anywhere in the kernel; six independent waves per CU (LDS-bound);
//   * lane (tile n = lane & 15, kq = lane >> 4) holds channels 16 j + 4 kq + e of k-step group j: one float4 of V feeds 4 (e) x CB MFMAs;
//     weights [f][j][cb][lane][4] go global -> registers one step ahead (4 KB per step; twice the L2 traffic per output of the
//     32-column form: the price of 16 columns per wave).
template <int CB>
__global__ __launch_bounds__(64, 2) void wino2x_kernel(const ConvArgs p, const int tiles_n, const int tgx, const int tgy, const int n_tg) {
    extern __shared__ v4f w2sm[];
    char* const halo = reinterpret_cast<char*>(w2sm);                      // [4 planes][5][5] pixels x 16 x 16 B
    const int lane = threadIdx.x;
    int blk;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, x = blockIdx.x & 7;
        blk = x * q + min(x, r) + (int)(blockIdx.x >> 3);                  // XCD-contiguous order
    }
    const int tile_n = blk % tiles_n, tg = blk / tiles_n;
    const int H = p.H, W = p.W;
    const int per_img = tgx * tgy;
    const int gn = tg / per_img, rem = tg - gn * per_img;
    const int gy = rem / tgx, gx = rem - gy * tgx;
    (void)n_tg;
    const int tn = lane & 15, kq = lane >> 4, tr = tn >> 2, tc = tn & 3;
    auto key = [](int pr, int pc) { return 2 * ((pc + 4 * (pr & 1)) & 7); };

    // ---- halo: 25 pieces of 4 LDS rows (= pixels) x 16 columns; lane -> row 4 i + (lane >> 4), physical column lane & 15
    {
        const int col = lane & 15;
        const float* const img = p.in + (size_t)gn * H * W * 64;
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            const int R = 4 * i + (lane >> 4);
            const int pl = R / 25, r2 = R - pl * 25, pr = r2 / 5, pc = r2 - pr * 5;
            const int y = 8 * gy - 1 + 2 * pr + (pl >> 1), x = 8 * gx - 1 + 2 * pc + (pl & 1);
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            dma16(ok ? img + ((size_t)y * W + x) * 64 + ((col ^ key(pr, pc)) << 2) : p.zeros, halo + i * 1024);
        }
    }
    typedef const __attribute__((address_space(3))) v4f* lds_v4f;
    const unsigned halo_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)halo;
    unsigned rb[16];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int pl = (dy & 1) * 2 + (dx & 1), pr = tr + (dy >> 1), pc = tc + (dx >> 1);
            rb[dy * 4 + dx] = halo_base + (unsigned)(((pl * 5 + pr) * 5 + pc) * 256 + ((key(pr, pc) ^ kq) << 4));
        }
    // weights of this column tile: step (f, j) = CB KB at ((f * 4 + j) * CB) * 256 floats; cb-th fragment + cb * 256
    // (a wave-uniform running pointer + the lane's 16-byte offset: with compile-time step offsets the 64 step addresses are hoisted out of
    //  the loop into 128 registers)
    const char* wstep = reinterpret_cast<const char*>(p.wt + (size_t)tile_n * 64 * CB * 256);
    const unsigned wlane = lane * 16;
    constexpr int WD = 4;                                                  // weight ring: fragments are requested WD - 1 steps (512 MFMA cycles each) ahead —
    v4f wr[WD][CB] = {};                                                   // one step ahead left an L2 round trip under load exposed at every step (-12 %)
#ifdef FACEHIP_W2_PROF
    const bool abl_w = p.sk_test_drop & 4, abl_d = p.sk_test_drop & 8, abl_y = p.sk_test_drop & 16, abl_st = p.sk_test_drop & 2;   // ablations (wrong results)
#else
    constexpr bool abl_w = false, abl_d = false, abl_y = false, abl_st = false;
#endif
    auto fetch_w = [&](bool advance, int buf) __attribute__((always_inline)) {
        if (abl_w) return;
        unsigned adv = advance ? CB * 1024 : 0;
        asm volatile("" : "+s"(adv));
        wstep += adv;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) wr[buf][cb] = *reinterpret_cast<const v4f*>(wstep + wlane + cb * 1024);
    };
    v4f Y[2][2][CB], Mp[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        Mp[cb] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) Y[a >> 1][a & 1][cb] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    auto y_update = [&](auto fc, const v4f (&M)[CB]) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value, i = f >> 2, j = f & 3;
        if (abl_y && f != 5) return;
        constexpr int c00 = kAT[0][i] * kAT[0][j], c01 = kAT[0][i] * kAT[1][j], c10 = kAT[1][i] * kAT[0][j], c11 = kAT[1][i] * kAT[1][j];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if constexpr (c00 > 0) Y[0][0][cb] += M[cb]; else if constexpr (c00 < 0) Y[0][0][cb] -= M[cb];
            if constexpr (c01 > 0) Y[0][1][cb] += M[cb]; else if constexpr (c01 < 0) Y[0][1][cb] -= M[cb];
            if constexpr (c10 > 0) Y[1][0][cb] += M[cb]; else if constexpr (c10 < 0) Y[1][0][cb] -= M[cb];
            if constexpr (c11 > 0) Y[1][1][cb] += M[cb]; else if constexpr (c11 < 0) Y[1][1][cb] -= M[cb];
            // pinned here: pure arithmetic floats freely in this barrier-free kernel, and left alone the scheduler defers the updates —
            // every deferred frequency keeps its 4 CB accumulator registers alive (256 registers + spills by the tenth frequency)
            if constexpr (c00 != 0) asm volatile("" : "+v"(Y[0][0][cb]));
            if constexpr (c01 != 0) asm volatile("" : "+v"(Y[0][1][cb]));
            if constexpr (c10 != 0) asm volatile("" : "+v"(Y[1][0][cb]));
            if constexpr (c11 != 0) asm volatile("" : "+v"(Y[1][1][cb]));
        }
    };
    v4f d[2][4] = {};
    auto fetch_d = [&](auto fc, auto jc, auto bc) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value, j = decltype(jc)::value, buf = decltype(bc)::value;
        constexpr int fi = f >> 2, fj = f & 3;
        constexpr int ya = kYA[fi], yb = kYB[fi], xa = kYA[fj], xb = kYB[fj];
        if (abl_d) return;
        unsigned gxr = j << 6;                                             // k-step group j = columns 4 j .. 4 j + 3 (opaque: see wino2_kernel;
        asm volatile("" : "+s"(gxr));                                      //  for j = 0 too: reads that depend on nothing are hoisted stages ahead)
        d[buf][0] = *reinterpret_cast<lds_v4f>((size_t)(rb[ya * 4 + xa] ^ gxr));
        d[buf][1] = *reinterpret_cast<lds_v4f>((size_t)(rb[ya * 4 + xb] ^ gxr));
        d[buf][2] = *reinterpret_cast<lds_v4f>((size_t)(rb[yb * 4 + xa] ^ gxr));
        d[buf][3] = *reinterpret_cast<lds_v4f>((size_t)(rb[yb * 4 + xb] ^ gxr));
    };
    auto stage = [&](auto fc) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value;
        constexpr int fi = f >> 2, fj = f & 3;
        constexpr int sab = kSA[fi] * kSB[fj], sba = kSB[fi] * kSA[fj], sbb = kSB[fi] * kSB[fj];
        v4f M[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) M[cb] = v4f{0.f, 0.f, 0.f, 0.f};
        auto step = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, k = f * 4 + j;
            fetch_w(k + WD - 1 < 64, (k + WD - 1) % WD);                    // fragments of step k + WD - 1 (past the end: a harmless re-fetch)
            if constexpr (j < 3) fetch_d(IC<f>{}, IC<j + 1>{}, IC<(k + 1) & 1>{});
            else if constexpr (f < 15) fetch_d(IC<f + 1>{}, IC<0>{}, IC<(k + 1) & 1>{});
            __builtin_amdgcn_sched_barrier(0);
            const v4f* dd = d[k & 1];
            v4f v = dd[0];
            v = sab > 0 ? v + dd[1] : v - dd[1];
            v = sba > 0 ? v + dd[2] : v - dd[2];
            v = sbb > 0 ? v + dd[3] : v - dd[3];
            asm volatile("" : "+v"(v));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) M[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[k % WD][cb][e], v[e], M[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        step(IC<0>{}); step(IC<1>{});
        y_update(IC<(f + 15) & 15>{}, Mp);                                 // the previous frequency's result -> the outputs, in the shadow of this one's MFMAs
        step(IC<2>{}); step(IC<3>{});
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) Mp[cb] = M[cb];
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch_w(false, 0);
#pragma unroll
    for (int q = 1; q < WD - 1; ++q) fetch_w(true, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the halo has landed: LDS-DMA is ordered for its own wave by vmcnt alone
    fetch_d(IC<0>{}, IC<0>{}, IC<0>{});
    stage(IC<0>{}); stage(IC<1>{}); stage(IC<2>{}); stage(IC<3>{}); stage(IC<4>{}); stage(IC<5>{}); stage(IC<6>{}); stage(IC<7>{});
    stage(IC<8>{}); stage(IC<9>{}); stage(IC<10>{}); stage(IC<11>{}); stage(IC<12>{}); stage(IC<13>{}); stage(IC<14>{}); stage(IC<15>{});
    y_update(IC<15>{}, Mp);

    // ---- epilogue: lane = tile (tr, tc), accumulator block cb = channels 16 cb + 4 kq .. + 3 of the column tile
    // (the kernel has no barrier, i.e. it is ONE basic block: without this fence the epilogue's bias / slope / residual loads are scheduled
    //  in front of the K loop and their ~80 destination registers are spilled through it)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (abl_st && lane != 77) return;
    const int oy0 = 2 * (4 * gy + tr), ox0 = 2 * (4 * gx + tc);
    const int n0 = tile_n * 16 * CB, Cout = p.Cout;
    const float* __restrict__ res = p.res;
    float* __restrict__ out1 = p.out1;
    float* __restrict__ out2 = p.out2;
    const bool relu = p.act == (int)Act::RELU, prelu = p.act == (int)Act::PRELU;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oy = oy0 + a, ox = ox0 + b;
            if (oy >= p.Ho || ox >= p.Wo) continue;
            const size_t pix = ((size_t)gn * p.Ho + oy) * p.Wo + ox;
            const int cls = p.bias_cls ? 3 * (oy == 0 ? 0 : oy == p.Ho - 1 ? 2 : 1) + (ox == 0 ? 0 : ox == p.Wo - 1 ? 2 : 1) : 0;
            if constexpr (CB == 2) {                                       // merged sibling convolutions (the only CB = 2 users): per-channel-range destination and activation
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = n0 + 16 * cb + 4 * kq + e;
                        if (co >= Cout) continue;
                        const int g = co >= p.oc0[2] && p.n_outs > 2 ? 2 : co >= p.oc0[1] ? 1 : 0;
                        const int cg = p.oc0[g + 1] - p.oc0[g];
                        float v = Y[a][b][cb][e] + p.bias[cls * Cout + co];
                        const int ga = p.oact[g];
                        v = ga == (int)Act::RELU ? fmaxf(v, 0.f) : ga == (int)Act::SIGMOID ? 1.0f / (1.0f + expf(-v)) : v;
                        p.outs[g][pix * cg + (co - p.oc0[g])] = v;
                    }
                continue;
            }
            if constexpr (CB == 2) continue;                               // (the vector epilogue below is the CB = 4 form's)
            const size_t row = pix * Cout + n0 + 4 * kq;
            v4f r4[CB];
            if (p.res_mode != (int)ResMode::NONE) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) r4[cb] = *reinterpret_cast<const v4f*>(res + row + 16 * cb);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int co = n0 + 16 * cb + 4 * kq;
                const v4f b4 = p.bias ? *reinterpret_cast<const v4f*>(p.bias + cls * Cout + co) : v4f{0.f, 0.f, 0.f, 0.f};
                const v4f sl = prelu ? *reinterpret_cast<const v4f*>(p.slope + co) : v4f{1.f, 1.f, 1.f, 1.f};
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act1(Y[a][b][cb][e] + b4[e], relu, sl[e]);
                if (p.res_mode != (int)ResMode::NONE) v += r4[cb];
                if (out1) *reinterpret_cast<v4f*>(out1 + row + 16 * cb) = v;
                if (out2) {
                    const v4f s2 = *reinterpret_cast<const v4f*>(p.s2 + co), t2 = *reinterpret_cast<const v4f*>(p.t2 + co);
                    *reinterpret_cast<v4f*>(out2 + row + 16 * cb) = v * s2 + t2;
                }
            }
        }
}

int pick_tgc(int W) {
    const int wt = (W + 1) / 2;                                            // tile columns of the map
    const int g7 = (wt + 6) / 7 * 8, g8 = (wt + 7) / 8 * 8;                // MFMA columns spent per tile row (a 7-wide group idles its eighth lane)
    return g7 < g8 ? 7 : 8;
}

int x16_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_WINO2_X16"); v = e ? atoi(e) : 1; }    // (0 = the 32-column kernel for 64-channel layers too: A / B timing)
    return v;
}
// layers the 16-column kernel takes: exactly 64 input channels; 64 k output channels, or <= 32 (SCRFD's merged head convolutions)
bool x16_shape(int Cin, int Cout) { return x16_mode() && Cin == 64 && (Cout % 64 == 0 || Cout <= 32); }   // (<= 32: merged outputs only, see wino2_ok)

template <int CB>
void launch_x16(const ConvArgs& a, hipStream_t s) {
    const int tgy = ((a.H + 1) / 2 + 3) / 4, tgx = ((a.W + 1) / 2 + 3) / 4;
    const long n_tg = (long)a.B * tgy * tgx;
    const int tiles_n = (a.Cout + 16 * CB - 1) / (16 * CB);
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL((wino2x_kernel<CB>), dim3((unsigned)(n_tg * tiles_n)), dim3(64), 25600, s, a, tiles_n, tgx, tgy, (int)n_tg);
    timer.end(s, 12, 2.0 * 16 * 16.0 * (double)n_tg * a.Cin * (16.0 * CB * tiles_n), a.t_flops);   // executed FLOPs (padded tiles / channels included)
}

template <int TGC>
void launch_tgc(const ConvArgs& a, hipStream_t s) {
    const int ht = (a.H + 1) / 2, wt = (a.W + 1) / 2;
    const int tgy = (ht + 3) / 4, tgx = (wt + TGC - 1) / TGC;
    const long n_tg = (long)a.B * tgy * tgx;
    const int tiles_n = a.Cout / 64;
    const long blocks = (n_tg + 1) / 2 * tiles_n;
    const size_t lds = Geo<TGC>::HALO_BYTES;
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
// 16-column kernel (Cin = 64): [column tile][16 f][4 j][CB blocks of 16 channels][64 lanes (m = lane & 15, kq = lane >> 4)][4 floats],
// element = U_f[cout = 16 CB tn + 16 cb + m][cin = 16 j + 4 kq + e]; channels >= Cout are zero rows.
size_t wino2_weight_floats(int Cin, int Cout) {
    if (x16_shape(Cin, Cout)) return (size_t)16 * Cin * (Cout <= 32 ? 32 : Cout);
    return (size_t)16 * Cin * Cout;
}
void wino2_pack_weights(const float* w, int Cout, int Cin, float* dst) {
    static const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    if (x16_shape(Cin, Cout)) {
        const int CB = Cout <= 32 ? 2 : 4;
        memset(dst, 0, wino2_weight_floats(Cin, Cout) * sizeof(float));
        for (int co = 0; co < Cout; ++co)
            for (int ci = 0; ci < Cin; ++ci) {
                double g[3][3], t[4][3];
                for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = w[((size_t)co * 9 + k) * Cin + ci];
                for (int i = 0; i < 4; ++i)
                    for (int x = 0; x < 3; ++x) t[i][x] = Gm[i][0] * g[0][x] + Gm[i][1] * g[1][x] + Gm[i][2] * g[2][x];
                const int tn = co / (16 * CB), cb = (co % (16 * CB)) / 16, m = co % 16;
                const int j = ci / 16, kq = (ci % 16) / 4, e = ci % 4;
                for (int i = 0; i < 4; ++i)
                    for (int jj = 0; jj < 4; ++jj) {
                        const double u = t[i][0] * Gm[jj][0] + t[i][1] * Gm[jj][1] + t[i][2] * Gm[jj][2];
                        const size_t step = ((size_t)tn * 16 + (i * 4 + jj)) * 4 + j;
                        dst[(step * CB + cb) * 256 + (kq * 16 + m) * 4 + e] = (float)u;
                    }
            }
        return;
    }
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
    if (!(a.ks == 3 && a.stride == 1 && a.pad == 1 && a.H == a.Ho && a.W == a.Wo && a.H >= 2 && a.W >= 2 && a.Cin % 32 == 0 && a.Cin >= 32 &&
          a.act != (int)Act::SIGMOID && !a.sc_in && !a.dw_w && !a.u8_src && a.wt_group_rows == 0 &&
          (a.res_mode == (int)ResMode::NONE || a.res_mode == (int)ResMode::SAME)))
        return false;
    if (a.n_outs > 0)                                                      // merged sibling convolutions: the 16-column kernel's epilogue only
        return x16_shape(a.Cin, a.Cout) && a.Cout <= 32 && a.res_mode == (int)ResMode::NONE && !a.out2 && !a.bias_cls;
    return a.Cout % 64 == 0;                                               // (plain layers: whole 64-channel column tiles)
}

// diagnostic builds (-DFACEHIP_W2_PROF, scripts/wino2_prof.sh): device buffer the kernel's phase stamps go to (1 MB, allocated on first use)
static float* g_w2_stamps = nullptr;
const void* wino2_stamp_buffer() {
    if (!g_w2_stamps) { void* q = nullptr; if (hipMalloc(&q, 1 << 20) == hipSuccess) { (void)hipMemset(q, 0, 1 << 20); g_w2_stamps = (float*)q; } }
    return g_w2_stamps;
}

long wino2_blocks(const ConvArgs& a) {
    if (x16_shape(a.Cin, a.Cout))                                          // one-wave workgroups: counted in units of four (a 256-thread workgroup's worth)
        return (long)a.B * (((a.H + 1) / 2 + 3) / 4) * (((a.W + 1) / 2 + 3) / 4) * ((a.Cout + 63) / 64) / 4;
    const int tgc = pick_tgc(a.W);
    const long n_tg = (long)a.B * (((a.H + 1) / 2 + 3) / 4) * (((a.W + 1) / 2 + tgc - 1) / tgc);
    return (n_tg + 1) / 2 * (a.Cout / 64);
}

// a.wt = wino2_pack_weights' image of the filter; everything else as for launch_conv
void launch_wino2(const ConvArgs& a_in, hipStream_t s) {
    if (!wino2_ok(a_in)) throw std::runtime_error("launch_wino2: layer shape not supported");
    ConvArgs a = a_in;
    a.zeros = conv_zero_line();
    a.slabs = g_w2_stamps;                                                 // (null unless a diagnostic run asked for the stamp buffer)
    { const char* e = getenv("FACEHIP_W2_ABLATE"); a.sk_test_drop = e ? atoi(e) : 0; }   // (read by diagnostic builds only)
    if (x16_shape(a.Cin, a.Cout)) {
        if (a.Cout <= 32) launch_x16<2>(a, s); else launch_x16<4>(a, s);
        return;
    }
    if (pick_tgc(a.W) == 7) launch_tgc<7>(a, s);
    else launch_tgc<8>(a, s);
}

}  // namespace fh
