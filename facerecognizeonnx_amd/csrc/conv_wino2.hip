// conv_wino2.hip — 3x3 stride-1 pad-1 convolutions with 64 input channels (IResNet-50's stage 1 and the stage-2 entry, SCRFD's merged
// 64 -> 30 head convolutions) as ONE fused Winograd F(2x2, 3x3) kernel on v_mfma_f32_16x16x4_f32 (gfx950 / CDNA4).
//
// What it replaces: Conv nodes inside `session_->Run` (reference src/face_recognizer.cpp:279-283, src/face_detector.cpp:179-183) —
// 112x112x64 -> 64, 4 x 56x56x64 -> 64, 56x56x64 -> 128: 2.7 ms of the 12.1 ms step at B = 128 in the direct form (conv_tall_kernel,
// 113-115 TFLOP/s = 0.73 of the f32 MFMA peak: at its ceiling).  Y = A^T [ (G g G^T) (.) (B^T d B) ] A on 2x2 output tiles needs 16
// multiplies per 4 outputs and input channel instead of 36: 2.25x less matrix-core work.  The unfused form (transform kernel -> GEMM ->
// transform kernel, as winograd.hip does for Cin >= 128) moves 4x the activation through memory twice around a K = 64 GEMM that is all
// prologue; here nothing but the layer's input and output touches memory:
//
//   * ONE WAVE per workgroup owns a 4 x 4 tile group (8 x 8 outputs = the 16 MFMA columns: lane = tile) and ALL 16 CB output channels
//     of its column tile (CB = 4: 64 channels; CB = 2: the <= 32 channels of merged sibling convolutions);
//   * the whole 64-channel halo of the group (10 x 10 pixels x 256 B = 25.6 KB) comes in once by LDS-DMA, de-interleaved into the four
//     (row parity, column parity) planes of 5 x 5 pixels — tile (tr, tc) reads patch pixel (dy, dx) from plane (dy & 1, dx & 1) at
//     pixel (tr + dy/2, tc + dx/2), so the lanes of one ds_read_b128 touch neighbouring rows of a [pixel][16 x float4] image whose
//     16-byte column is XOR-swizzled by 2 ((pc + 4 (pr & 1)) & 7): conflict-free (scripts/wino2_banks.py enumerates every read; with
//     interleaved pixels every lane's address has the same parity and no swizzle avoids a 2-way conflict).  A wave only reads what it
//     loaded itself: NO barrier anywhere in the kernel; six independent waves per CU (LDS-bound);
//   * for each of the 16 frequencies f = (i, j) in turn: lane (tile n = lane & 15, kq = lane >> 4) forms V_f = (B^T d B)[i][j] of ITS
//     tile for channels 16 g + 4 kq + e of k-step group g from four ds_read_b128 and three adds (B^T has two +-1 entries per row) — a
//     float4 that is exactly the B fragment the MFMA wants from that lane and feeds 4 (e) x CB MFMAs; U_f = G g G^T (fp64 at load time,
//     carrying the block's folded BatchNorm) is pre-arranged [f][g][cb][lane][4] and goes global -> registers a few steps ahead;
//     M_f accumulates in 4 CB registers and is added into the four output accumulators Y[a][b] with A^T's {0, +-1} coefficients (1, 2 or
//     4 adds per register) in the shadow of the NEXT frequency's MFMAs;
//   * epilogue on the 2x2 pixels of the lane's tile: bias (9 border classes when the block's BatchNorm is folded in, engine.cpp) ->
//     ReLU / PReLU -> + residual -> store (+ second output); or, CB = 2, per-channel-range destinations and activations (merged heads).
//
// A 32-column form of the same algorithm (v_mfma_f32_32x32x2_f32: 4 x 7 | 8 tiles per wave, two tile groups x two 32-channel halves per
// 4-wave workgroup, 32-channel halo chunks, weights through an LDS stage or registers, persistent or not) was built first and measured
// equal: 185-197 us on 56x56x64 -> 64 at B = 128 for every variant, against 285 us direct (docs/kernels.md 3.1l has the ablation that says why:
// the costs ADD — the kernel behaves like one bound by energy, not by any one pipe).  This form is kept: exact fit on 56 / 112 / 80 / 40-wide
// maps, V formed once per tile instead of once per channel half, no barrier, and the merged-output epilogue.
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

// B^T of F(2x2, 3x3): row i has two non-zero entries, at patch rows YA[i] (sign +1) and YB[i] (sign SB[i])
//   [ 1  0 -1  0 ]   [ 0  1  1  0 ]   [ 0 -1  1  0 ]   [ 0  1  0 -1 ]
constexpr int kYA[4] = {0, 1, 2, 1}, kYB[4] = {2, 2, 1, 3};
constexpr int kSA[4] = {1, 1, 1, 1}, kSB[4] = {-1, 1, -1, -1};
// A^T = [ 1 1 1 0 ; 0 1 -1 -1 ]: coefficient of frequency row i in output row a
constexpr int kAT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};

template <int V> struct IC { static constexpr int value = V; };          // compile-time integers: the loop over the 16 frequencies is a fold over
                                                                          // 16 instantiations (`#pragma unroll` gives up on a body of this size, and a
                                                                          // run-time f turns every B^T / A^T coefficient into a branch)

// NW = waves per workgroup that share ONE halo: wave w computes the 16 CB channels [16 CB w, 16 CB (w + 1)) of the workgroup's 16 CB NW-channel
// column tile (NW = 1: the original one-wave form).  MERGED selects the merged-sibling epilogue (per-channel-range destinations).
// NW = 2, CB = 2 on the plain 64-channel layers: the 25.6 KB halo limits a CU to six workgroups either way, so six waves of 226 registers
// (1.5 per SIMD: two SIMDs carry two waves, two carry one) become twelve of 112 (three per SIMD) — DESIGN.md 3.1.
template <int CB, int NW, bool MERGED>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 2 : 3) void wino2_kernel(const ConvArgs p, const int tiles_n, const int tgx, const int tgy, const int n_tg) {
    extern __shared__ v4f w2sm[];
    char* const halo = reinterpret_cast<char*>(w2sm);                      // [5][5] plane pixels x [4 parity planes] x 16 x 16 B
    const int lane = threadIdx.x & 63;
    const int wv = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    constexpr int CBT = CB * NW;                                           // 16-channel blocks per column tile in the weight image
#ifdef FACEHIP_W2_PROF
    const long long prof_c0 = __builtin_readcyclecounter(), prof_r0 = __builtin_amdgcn_s_memrealtime();   // shader clock vs the constant 100 MHz counter
#endif
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

    // ---- halo: 25 pieces of 4 LDS rows x 16 columns.  LDS row 4 i + pl holds pixel i = 5 pr + pc of parity plane pl, so piece i is the same
    // plane pixel of the four planes: its lanes' plane is lane >> 4 (computed once), its (pr, pc) and swizzle key are compile-time constants
    // — ~10 VALU per piece instead of ~30 with plane-major rows (the bank a read hits does not depend on its row: a row is all 64 banks)
    {
        const int col = lane & 15, pl = lane >> 4;
        const int y0 = 8 * gy - 1 + (pl >> 1), x0 = 8 * gx - 1 + (pl & 1);
        const float* const img = p.in + ((size_t)gn * H * W + (size_t)y0 * W + x0) * 64;   // (may point in front of the image: only dereferenced when in range)
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            const int pr = i / 5, pc = i - 5 * pr;
            const int y = y0 + 2 * pr, x = x0 + 2 * pc;
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            if (NW > 1 && (i % NW) != wv) continue;                        // (wave-uniform: the waves of a workgroup take alternate pieces)
            dma16(ok ? img + ((long)2 * pr * W + 2 * pc) * 64 + ((col ^ key(pr, pc)) << 2) : p.zeros, halo + i * 1024);
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
            rb[dy * 4 + dx] = halo_base + (unsigned)((4 * (pr * 5 + pc) + pl) * 256 + ((key(pr, pc) ^ kq) << 4));
        }
    // weights of this column tile: step (f, j) = CB KB at ((f * 4 + j) * CB) * 256 floats; cb-th fragment + cb * 256
    // (a wave-uniform running pointer + the lane's 16-byte offset: with compile-time step offsets the 64 step addresses are hoisted out of
    //  the loop into 128 registers)
    const char* wstep = reinterpret_cast<const char*>(p.wt + ((size_t)tile_n * 64 * CBT + (size_t)wv * CB) * 256);
    const unsigned wlane = lane * 16;
    constexpr int WD = 2;                                                  // weight ring: fragments are requested WD - 1 steps (512 MFMA cycles each) ahead
    v4f wr[WD][CB] = {};                                                   // (WD = 4 measured the same: 196.0 vs 194.6 us on 56x56x64, and costs 32 registers)
#ifdef FACEHIP_W2_PROF
    const bool abl_w = p.sk_test_drop & 4, abl_d = p.sk_test_drop & 8, abl_y = p.sk_test_drop & 16, abl_st = p.sk_test_drop & 2;   // ablations (wrong results)
#else
    constexpr bool abl_w = false, abl_d = false, abl_y = false, abl_st = false;
#endif
    auto fetch_w = [&](bool advance, int buf) __attribute__((always_inline)) {
        if (abl_w) return;
        unsigned adv = advance ? CBT * 1024 : 0;
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
        unsigned gxr = j << 6;                                             // k-step group j = columns 4 j .. 4 j + 3 (an opaque scalar: a compile-time
        asm volatile("" : "+s"(gxr));                                      //  XOR is hoisted — 48 addresses in registers; for j = 0 too: reads that depend on nothing float stages ahead)
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
    if constexpr (NW > 1) __syncthreads();                                 // ... and for the sibling wave's pieces by the workgroup's only barrier
    fetch_d(IC<0>{}, IC<0>{}, IC<0>{});
    stage(IC<0>{}); stage(IC<1>{}); stage(IC<2>{}); stage(IC<3>{}); stage(IC<4>{}); stage(IC<5>{}); stage(IC<6>{}); stage(IC<7>{});
    stage(IC<8>{}); stage(IC<9>{}); stage(IC<10>{}); stage(IC<11>{}); stage(IC<12>{}); stage(IC<13>{}); stage(IC<14>{}); stage(IC<15>{});
    y_update(IC<15>{}, Mp);

    // ---- epilogue: lane = tile (tr, tc), accumulator block cb = channels 16 cb + 4 kq .. + 3 of the column tile
    // (the kernel has no barrier, i.e. it is ONE basic block: without this fence the epilogue's bias / slope / residual loads are scheduled
    //  in front of the K loop and their ~80 destination registers are spilled through it)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (abl_st && lane != 0) return;                                       // (lane 0 goes on to the clock stamp; 63 of 64 lanes' stores are gone)
    const int oy0 = 2 * (4 * gy + tr), ox0 = 2 * (4 * gx + tc);
    const int n0 = (tile_n * NW + wv) * 16 * CB, Cout = p.Cout;
    const float* __restrict__ res = p.res;
    float* __restrict__ out1 = p.out1;
    float* __restrict__ out2 = p.out2;
    const bool relu = p.act == (int)Act::RELU, prelu = p.act == (int)Act::PRELU;
    float bm[CB][4];                                                       // merged form: this lane's 8 bias values, fetched once (no load between the stores)
    if constexpr (MERGED) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int co = n0 + 16 * cb + 4 * kq + e; bm[cb][e] = co < Cout ? p.bias[co] : 0.f; }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oy = oy0 + a, ox = ox0 + b;
            if (!MERGED || oy >= p.Ho || ox >= p.Wo) continue;
            const size_t pix = ((size_t)gn * p.Ho + oy) * p.Wo + ox;       // (merged convolutions carry no folded BatchNorm: wino2_ok)
            if constexpr (MERGED) {                                        // merged sibling convolutions: per-channel-range destination and activation
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {                       // channel pairs: one 8-byte store where both land in one destination, aligned
                        const int co = n0 + 16 * cb + 4 * kq + e;
                        if (co >= Cout) continue;
                        const int g = co >= p.oc0[2] && p.n_outs > 2 ? 2 : co >= p.oc0[1] ? 1 : 0;
                        const int cg = p.oc0[g + 1] - p.oc0[g], ga = p.oact[g];
                        auto fin = [&](float v) { return ga == (int)Act::RELU ? fmaxf(v, 0.f) : ga == (int)Act::SIGMOID ? 1.0f / (1.0f + expf(-v)) : v; };
                        float* const dst = p.outs[g] + pix * cg + (co - p.oc0[g]);
                        const float v0 = fin(Y[a][b][cb][e] + bm[cb][e]);
                        if (co + 1 < p.oc0[g + 1] && !((cg | (co - p.oc0[g])) & 1)) {
                            *reinterpret_cast<float2*>(dst) = float2{v0, fin(Y[a][b][cb][e + 1] + bm[cb][e + 1])};
                        } else {
                            dst[0] = v0;
                            if (co + 1 < Cout) {                          // the pair straddles two destinations (or is unaligned): scalar stores
                                const int g1 = co + 1 >= p.oc0[2] && p.n_outs > 2 ? 2 : co + 1 >= p.oc0[1] ? 1 : 0;
                                const int cg1 = p.oc0[g1 + 1] - p.oc0[g1], ga1 = p.oact[g1];
                                float v1 = Y[a][b][cb][e + 1] + bm[cb][e + 1];
                                v1 = ga1 == (int)Act::RELU ? fmaxf(v1, 0.f) : ga1 == (int)Act::SIGMOID ? 1.0f / (1.0f + expf(-v1)) : v1;
                                p.outs[g1][pix * cg1 + (co + 1 - p.oc0[g1])] = v1;
                            }
                        }
                    }
                continue;
            }
        }
    if constexpr (!MERGED) {
        // Every load of the epilogue goes out BEFORE the first store: bias per pixel (its border class), slope, residual — 36 float4 per lane
        // in flight together, ONE global-memory latency.  (Per pixel "load, wait, compute, store" was four latencies in a row: with the
        // stores masked to one lane of 64 the epilogue still cost 33 of the layer's 195 us — it was never the traffic.)
        const int n0 = (tile_n * NW + wv) * 16 * CB + 4 * kq;
        bool live[4]; size_t row[4]; v4f b4[4][CB], r4[4][CB], sl[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) sl[cb] = prelu ? *reinterpret_cast<const v4f*>(p.slope + n0 + 16 * cb) : v4f{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = oy0 + (q >> 1), ox = ox0 + (q & 1);
            live[q] = oy < p.Ho && ox < p.Wo;
            const int oyc = min(oy, p.Ho - 1), oxc = min(ox, p.Wo - 1);     // (clamped: dead pixels load a neighbour's vectors and store nothing)
            row[q] = (((size_t)gn * p.Ho + oyc) * p.Wo + oxc) * Cout + n0;
            const int cls = p.bias_cls ? 3 * (oyc == 0 ? 0 : oyc == p.Ho - 1 ? 2 : 1) + (oxc == 0 ? 0 : oxc == p.Wo - 1 ? 2 : 1) : 0;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                b4[q][cb] = p.bias ? *reinterpret_cast<const v4f*>(p.bias + cls * Cout + n0 + 16 * cb) : v4f{0.f, 0.f, 0.f, 0.f};
                if (p.res_mode != (int)ResMode::NONE) r4[q][cb] = *reinterpret_cast<const v4f*>(res + row[q] + 16 * cb);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!live[q]) continue;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act1(Y[q >> 1][q & 1][cb][e] + b4[q][cb][e], relu, sl[cb][e]);
                if (p.res_mode != (int)ResMode::NONE) v += r4[q][cb];
                if (out1) *reinterpret_cast<v4f*>(out1 + row[q] + 16 * cb) = v;
                if (out2) {
                    const v4f s2 = *reinterpret_cast<const v4f*>(p.s2 + n0 + 16 * cb), t2 = *reinterpret_cast<const v4f*>(p.t2 + n0 + 16 * cb);
                    *reinterpret_cast<v4f*>(out2 + row[q] + 16 * cb) = v * s2 + t2;
                }
            }
        }
    }
#ifdef FACEHIP_W2_PROF
    if (lane == 0 && p.slabs && blockIdx.x < (1u << 17)) {                 // this wave's mean shader clock in MHz x 10 -> the diagnostic buffer (never an output)
        const long long dc = __builtin_readcyclecounter() - prof_c0, dr = (long long)__builtin_amdgcn_s_memrealtime() - prof_r0;
        reinterpret_cast<float*>(p.slabs)[blockIdx.x] = (float)(dc * 1000 / (dr + 1));
    }
#endif
}

bool shape_ok(int Cin, int Cout) { return Cin == 64 && (Cout % 64 == 0 || Cout <= 32); }   // (<= 32: merged outputs only, see wino2_ok)

template <int CB, int NW, bool MERGED>
void launch_cb(const ConvArgs& a, hipStream_t s) {
    const int tgy = ((a.H + 1) / 2 + 3) / 4, tgx = ((a.W + 1) / 2 + 3) / 4;
    const long n_tg = (long)a.B * tgy * tgx;
    const int tiles_n = (a.Cout + 16 * CB * NW - 1) / (16 * CB * NW);      // column tiles of 16 CB NW channels: one workgroup each
    // The kernel XORs swizzle keys into ABSOLUTE LDS addresses (rb[] ^ (j << 6)): that equals base + (offset ^ key) only while the dynamic
    // LDS base is 256-byte aligned, i.e. 0 — true as long as the kernel declares no static __shared__ object.  Checked once per instantiation.
    static const bool lds_ok = [] {
        hipFuncAttributes fa{};
        FH_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&wino2_kernel<CB, NW, MERGED>)));
        return fa.sharedSizeBytes % 256 == 0;
    }();
    if (!lds_ok) throw std::runtime_error("wino2: static LDS moved the halo off its 256-byte alignment (see the address XOR in fetch_d)");
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL((wino2_kernel<CB, NW, MERGED>), dim3((unsigned)(n_tg * tiles_n)), dim3(64 * NW), 25600, s, a, tiles_n, tgx, tgy, (int)n_tg);
    // booked with the FLOPs the matrix cores EXECUTE (padded tiles / channels included); bytes slot: the layer's direct-form FLOPs, as tag 7
    timer.end(s, 12, 2.0 * 16 * 16.0 * (double)n_tg * a.Cin * (16.0 * CB * NW * tiles_n), a.t_flops);
}

}  // namespace

// U_f = G g G^T per (output channel, input channel) in fp64, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], in the order the kernel streams it:
// [column tile][16 f][4 g][CB blocks of 16 channels][64 lanes (m = lane & 15, kq = lane >> 4)][4 floats],
// element = U_f[cout = 16 CB tn + 16 cb + m][cin = 16 g + 4 kq + e]; channels >= Cout are zero rows.  w = [Cout][9 taps][Cin] (the engine's layout).
size_t wino2_weight_floats(int Cin, int Cout) { return (size_t)16 * Cin * (Cout <= 32 ? 32 : Cout); }
void wino2_pack_weights(const float* w, int Cout, int Cin, float* dst) {
    static const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    if (!shape_ok(Cin, Cout)) throw std::runtime_error("wino2_pack_weights: layer shape not supported");
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
}

bool wino2_ok(const ConvArgs& a) {
    if (!(a.ks == 3 && a.stride == 1 && a.pad == 1 && a.H == a.Ho && a.W == a.Wo && a.H >= 2 && a.W >= 2 && shape_ok(a.Cin, a.Cout) &&
          a.act != (int)Act::SIGMOID && !a.sc_in && !a.dw_w && !a.u8_src && a.wt_group_rows == 0 &&
          (a.res_mode == (int)ResMode::NONE || a.res_mode == (int)ResMode::SAME)))
        return false;
    if (a.n_outs > 0)                                                      // merged sibling convolutions: per-range destinations, no residual / second output
        return a.Cout <= 32 && a.res_mode == (int)ResMode::NONE && !a.out2 && !a.bias_cls;
    return a.Cout % 64 == 0;                                               // plain layers: whole 64-channel column tiles
}

// diagnostic runs (FACEHIP_W2_ABLATE set, diagnostic build): per-workgroup shader-clock stamps land in a 512 KB device buffer; returns their
// median in MHz (0 in production builds, which never write it)
static float* g_w2_clock = nullptr;
double wino2_debug_clock_mhz() {
    if (!g_w2_clock) return 0.0;
    std::vector<float> h(1 << 17);
    if (hipMemcpy(h.data(), g_w2_clock, h.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
    std::vector<float> v;
    for (float x : h) if (x > 0.f) v.push_back(x);
    if (v.empty()) return 0.0;
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    return v[v.size() / 2] / 10.0;
}

// one-wave workgroups of the launch, in units of four (= the 256-thread workgroups the engine's cross-over is expressed in)
long wino2_blocks(const ConvArgs& a) {
    return (long)a.B * (((a.H + 1) / 2 + 3) / 4) * (((a.W + 1) / 2 + 3) / 4) * ((a.Cout + 63) / 64) / 4;
}

// a.wt = wino2_pack_weights' image of the filter; everything else as for launch_conv
void launch_wino2(const ConvArgs& a_in, hipStream_t s) {
    if (!wino2_ok(a_in)) throw std::runtime_error("launch_wino2: layer shape not supported");
    ConvArgs a = a_in;
    a.zeros = conv_zero_line();
    static const int ablate = [] { const char* e = getenv("FACEHIP_W2_ABLATE"); return e ? atoi(e) : -1; }();
    a.sk_test_drop = ablate < 0 ? 0 : ablate;                              // (read by diagnostic builds only: scripts/wino2_prof.sh)
    if (ablate >= 0 && !g_w2_clock) { void* q = nullptr; if (hipMalloc(&q, sizeof(float) << 17) == hipSuccess) g_w2_clock = (float*)q; }
    if (g_w2_clock) (void)hipMemsetAsync(g_w2_clock, 0, sizeof(float) << 17, s);
    a.slabs = g_w2_clock;
    // plain 64-channel column tiles: two waves of 32 channels around one halo (FACEHIP_WINO2_NW=1: the one-wave form of round 4, for A / B timing)
    static const int nw = [] { const char* e = getenv("FACEHIP_WINO2_NW"); return e ? atoi(e) : 2; }();
    if (a.Cout <= 32) launch_cb<2, 1, true>(a, s);
    else if (nw == 2) launch_cb<2, 2, false>(a, s);
    else launch_cb<4, 1, false>(a, s);
}

}  // namespace fh
