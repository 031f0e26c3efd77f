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
// Tile anatomy (256 threads = 4 waves, >= 2 workgroups per CU):
//   * global -> LDS directly (LDS-DMA, global_load_lds_dwordx4: no staging registers, no
//     ds_write), double buffered; one barrier per 32-deep K chunk.  Padding, ragged M and ragged
//     K are resolved by redirecting the lane's source address to a zero line (no branches).
//   * LDS image [row][32 k] with the 16-byte column XOR-swizzled by (row>>1)&7, which makes the
//     ds_read_b128 fragment reads of 32 different rows conflict-free (MI355X_MICROARCH.md §LDS).
//   * each lane fetches 4 consecutive k of its row with ONE ds_read_b128 and feeds 4 MFMAs; the
//     k-order inside a chunk is therefore permuted identically for both operands, which a dot
//     product does not care about.  A wave tile of 64x64 needs 4 ds_read_b128 per 16 MFMAs;
//     the fragments of the next 8-deep step are fetched while the current 16 MFMAs issue.
//   * the MFMA's A operand is the WEIGHT fragment and B the PIXEL fragment, so a lane ends up
//     with one pixel and, per accumulator quad, 4 consecutive output channels: the epilogue
//     moves float4s (residual in, output out, second output out).
//   * epilogue: bias -> activation -> (+ residual, optionally through a 2x nearest up-sampling)
//     -> store, plus an optional second output  y*s2 + t2  (the next block's pre-conv BN).
//
// Work distribution ("stream-K remainder"): with T tiles and S workgroups resident on the chip,
// the first floor(T/S)*S tiles are computed one per workgroup.  The remaining R < S tiles would
// leave S-R slots idle, so their K range is cut: R OWNER workgroups compute the first q_o chunks
// of their tile, H <= S-R HELPER workgroups (lower block indices, so they are dispatched first)
// share the other chunks evenly, each covering a contiguous run that may span several tiles.
// A helper stores its raw accumulators to a slab and publishes it (agent-scope release + one
// counter add per tile); the owner polls its tile's counter, acquires, adds the <= 3 slabs in K
// order and runs the epilogue — deterministic, no float atomics, no second kernel.  When a tile
// would need more than 3 slabs (the 25088-deep FC as split-K, tiny remainders) there are no
// owners: every segment goes to a slab and conv_fixup_kernel sums them in a second launch.
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <vector>

#include "kernels.h"
#include "plan.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// stream-K workspace: accumulator slabs followed by one counter word per remainder tile
constexpr size_t SK_SLAB_FLOATS = (size_t)2 * 1280 * 128 * 128;
constexpr size_t SK_COUNTERS = 4096;

// The same with the activation a compile-time constant: the epilogues' hot paths switch ONCE per tile (with_act) instead of running
// apply_act's branch tree per element — inlined 16 times per 32 x 32 block it made the epilogue thousands of basic blocks
// (phase stamps of conv_pw_kernel, scripts/pw_ablate.sh: first block in LDS -> last store issued 7.7 -> 5.8 us of a 37 us tile).
template <int A>
__device__ __forceinline__ float act_c(float v, float slope) {
    if constexpr (A == (int)Act::RELU) return v > 0.f ? v : 0.f;
    else if constexpr (A == (int)Act::PRELU) return v >= 0.f ? v : v * slope;
    else if constexpr (A == (int)Act::SIGMOID) return 1.0f / (1.0f + expf(-v));
    else return v;
}
template <int A> struct ActC { static constexpr int value = A; };
template <class F>
__device__ __forceinline__ void with_act(int act, F&& f) {
    if (act == (int)Act::RELU) f(ActC<(int)Act::RELU>{});
    else if (act == (int)Act::PRELU) f(ActC<(int)Act::PRELU>{});
    else if (act == (int)Act::SIGMOID) f(ActC<(int)Act::SIGMOID>{});
    else f(ActC<(int)Act::NONE>{});
}
// ds_write_b128 the compiler does not see (p points into LDS; ordered before later LDS accesses of the wave by the LDS queue itself)
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == (int)Act::RELU) return v > 0.f ? v : 0.f;
    if (act == (int)Act::PRELU) return v >= 0.f ? v : v * slope;
    if (act == (int)Act::SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// global -> LDS without a register round trip (global_load_lds_dwordx4).  The global address is
// per lane; the LDS address is the wave-uniform `dst` + 16 * lane.  (The builtin only exists in the
// device pass; the host pass of hipcc just needs the kernel body to parse.)
__device__ __forceinline__ void lds_dma16(const float* src, v4f* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

template <int BM, int BN, int WM, int WN>
struct Tile {
    static constexpr int T = WM * WN * 64;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int RP = T / 8;                    // tile rows filled per loader pass
    static constexpr int AL = BM / RP, BL = BN / RP;
    static constexpr int SLAB = BM * BN;                // floats per accumulator slab
    static_assert(TM >= 1 && TN >= 1 && AL >= 1 && BL >= 1 && RP % 16 == 0, "tile shape");
};

// ---- epilogue shared by the main kernel and the fix-up kernel -------------------------------
// C/D map of the 32x32 MFMA: column (= pixel here) = lane & 31, row (= channel) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// ep (optional): the tile's per-channel vectors in LDS, float [12][BN]: rows 0..8 = bias (one per border class when bias_cls, else row 0),
// 9 = PReLU slope, 10 / 11 = s2 / t2 of the second output; channels >= Cout hold 0.  With it the epilogue issues NO global load between
// its stores: on this ISA loads and stores share one in-order counter (vmcnt), so a load issued after a store can only be waited for
// together with that store's acknowledgement — bias / slope loads interleaved with the stores turned the epilogue into a chain of
// store round trips (16 per 256x64 tile, a third of the tile's time).  The residual reads are issued up front for the same reason.
// whole-line epilogue (conv_epilogue: tr) when rows are whole 128-byte lines; ep_direct = A / B switch (FACEHIP_EP_DIRECT): 1 = never,
// 3 = for every Cout % 4 == 0 (measured: the 16-channel FPN laterals lose 5 % to their idle tail lanes)
__device__ __forceinline__ bool conv_ep_lines(const ConvArgs& p) { return p.ep_direct == 0 ? (p.Cout & 31) == 0 : p.ep_direct == 3; }
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, v16f (&acc)[BM / WM / 32][BN / WN / 32], int m0, int n0,
                                              int wm, int wn, int lane, int only_i = -1, int only_j = -1, const float* ep = nullptr,
                                              float* tr = nullptr) {
    using TL = Tile<BM, BN, WM, WN>;
    const int fr = lane & 31, fh2 = lane >> 5;
    const int HoWo = p.Ho * p.Wo;
    const int M = p.B * HoWo;
    const float* __restrict__ res = p.res;
    float* __restrict__ out1 = p.out1;
    float* __restrict__ out2 = p.out2;
    const bool vec = (p.Cout & 3) == 0;
    if (p.n_outs > 0) {                                  // merged sibling convs: per-channel-range destination
#pragma unroll
        for (int i = 0; i < TL::TM; ++i) {
            const int m = m0 + (wm * TL::TM + i) * 32 + fr;
            if (m >= M || (only_i >= 0 && i != only_i)) continue;
#pragma unroll
            for (int j = 0; j < TL::TN; ++j) {
                if (only_j >= 0 && j != only_j) continue;
                const int cb = n0 + (wn * TL::TN + j) * 32 + 4 * fh2;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = cb + 8 * (e >> 2) + (e & 3);
                    if (co >= p.Cout) continue;
                    const int g = co >= p.oc0[2] && p.n_outs > 2 ? 2 : co >= p.oc0[1] ? 1 : 0;
                    const int cg = p.oc0[g + 1] - p.oc0[g];
                    const float v = apply_act(acc[i][j][e] + p.bias[co], p.oact[g], 0.f);
                    p.outs[g][(size_t)m * cg + (co - p.oc0[g])] = v;
                }
            }
        }
        return;
    }
    if (ep && vec && tr && only_i < 0 && only_j < 0) {
        // Whole-line form (round 4).  A lane's accumulators are 4-channel pieces of ITS pixel row: written straight from registers, one
        // store instruction puts 32 bytes into each of 32 rows — a quarter of a 128-byte line per row, 8 instructions (and 8 residual
        // reads) per line.  Here each 32 x 32 block is turned around in a wave-private LDS scratch (tr: [waves][32 rows][36 floats], the
        // 16 lanes of a b128 phase hit different banks on the write side) and a lane takes float4 column cq of rows rr, rr + 8, rr + 16,
        // rr + 24: one instruction then moves 8 rows x ONE whole line, for stores and residual reads alike, and the per-channel vectors of
        // a block are the same for all four of a lane's rows (whole lines when Cout % 32 == 0, whole 128-byte runs otherwise).  Same arithmetic
        // per element: bit-identical results.
        float* const blk = tr + (wm * WN + wn) * (32 * 36);
        const int rr = lane >> 3, cq = lane & 7;
#ifdef FACEHIP_PW_ABL
        const unsigned long long est0 = wall_clock64();
        unsigned long long est1 = 0, est2 = 0, est3 = 0;
#endif
        // Loads and stores share ONE in-order counter (vmcnt): a wait for a residual load that sits behind a store in program order is a
        // wait for that store's acknowledgement (~0.36 us; with the residual a run-time flag the compiler kept such a wait in front of every
        // row group even when there was no residual).  So the residual is a compile-time flag here, a block's residual reads are issued
        // first and consumed two row groups at a time BEFORE those groups' stores.  (What remains between blocks is the compiler's
        // vmcnt(0) in front of the next block's LDS stores — it protects the registers the pending global stores read; giving every stored
        // value a register of its own removed it and changed nothing: the tile's 48 KB leave at the chip's ~5 TB/s write rate together
        // with every other workgroup's, scripts/pw_phases.py.)
        auto body = [&](auto AC, auto RC) {
        constexpr int A = decltype(AC)::value;
        constexpr bool R = decltype(RC)::value != 0;
#pragma unroll
        for (int i = 0; i < TL::TM; ++i) {
            const int mb = m0 + (wm * TL::TM + i) * 32 + rr;               // this lane's rows: mb + 8 k
            int cls4 = 0;                                                   // border class of row k in bits 4k .. 4k+3
            if (p.bias_cls) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int mc = min(mb + 8 * k, M - 1);
                    const int n = mc / HoWo, rem = mc - n * HoWo;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    cls4 |= (3 * (oy == 0 ? 0 : oy == p.Ho - 1 ? 2 : 1) + (ox == 0 ? 0 : ox == p.Wo - 1 ? 2 : 1)) << (4 * k);
                }
            }
            auto res_row = [&](int m) -> size_t {
                if (p.res_mode != (int)ResMode::UP2X) return (size_t)m * p.Cout;
                const int n = m / HoWo, rem = m - n * HoWo;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                return ((size_t)(n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout;
            };
#pragma unroll
            for (int j = 0; j < TL::TN; ++j) {
                const int cl = (wn * TL::TN + j) * 32 + 4 * cq;            // channel within the tile
                const int co = n0 + cl;
                if (n0 + (wn * TL::TN + j) * 32 >= p.Cout) continue;       // (a padded column block; wave-uniform)
                const bool cok = co < p.Cout;                              // (the last block of a Cout that is not a multiple of 32: its tail lanes idle)
                v4f r4[4];
                if constexpr (R) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) r4[k] = cok ? *reinterpret_cast<const v4f*>(res + res_row(min(mb + 8 * k, M - 1)) + co) : v4f{0.f, 0.f, 0.f, 0.f};
                }
                wave_lds_order();                                            // (the previous block's scratch reads lie above these writes)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<v4f*>(blk + fr * 36 + 8 * g + 4 * fh2) = v4f{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                wave_lds_order();                                            // (lane (rr, cq) reads rows other lanes wrote)
                v4f sl = {0.f, 0.f, 0.f, 0.f}, s2 = sl, t2 = sl;
                if constexpr (A == (int)Act::PRELU) sl = *reinterpret_cast<const v4f*>(ep + 9 * BN + cl);
                if (out2) { s2 = *reinterpret_cast<const v4f*>(ep + 10 * BN + cl); t2 = *reinterpret_cast<const v4f*>(ep + 11 * BN + cl); }
#ifdef FACEHIP_PW_ABL
                if (i == 0 && j == 0) est1 = wall_clock64();                // first block's accumulators written to LDS (issued)
#endif
                // (two row groups at a time: their arithmetic, then their stores — four at a time spill in the 128-register instantiations)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v4f vk[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int k = 2 * h + q;
                        const v4f a4 = *reinterpret_cast<const v4f*>(blk + (rr + 8 * k) * 36 + 4 * cq);
                        const v4f b4 = *reinterpret_cast<const v4f*>(ep + ((cls4 >> (4 * k)) & 15) * BN + cl);
#pragma unroll
                        for (int c = 0; c < 4; ++c) vk[q][c] = act_c<A>(a4[c] + b4[c], sl[c]);
                        if constexpr (R) vk[q] += r4[k];
                    }
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);                      // (every wait of this half lies above its stores)
#endif
#ifdef FACEHIP_PW_ABL
                    if (i == 0 && j == 0 && h == 0) est2 = wall_clock64();  // first block's first values ready
#endif
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int m = mb + 8 * (2 * h + q);
#ifdef FACEHIP_PW_ABL
                        if ((p.sk_test_drop & 16) && lane != 0) continue;  // (diagnostic: every instruction of the epilogue but 63 of 64 lanes' stores)
#endif
                        if (m < M && cok) {
                            const size_t row = (size_t)m * p.Cout + co;
                            if (out1) *reinterpret_cast<v4f*>(out1 + row) = vk[q];
                            if (out2) *reinterpret_cast<v4f*>(out2 + row) = vk[q] * s2 + t2;
                        }
                    }
                }
#ifdef FACEHIP_PW_ABL
                if (i == 0 && j == 0) est3 = wall_clock64();                // first block's four stores issued
#endif
            }
        }
        };
        with_act(p.act, [&](auto AC) {
            if (p.res_mode != (int)ResMode::NONE) body(AC, ActC<1>{}); else body(AC, ActC<0>{});
        });
#ifdef FACEHIP_PW_ABL
        if (threadIdx.x == 0 && p.slabs && p.Cout == 288 && p.Cin == 288) {   // (phase stamps of conv_pw_kernel, second half: see there)
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.slabs) + 8192 + (size_t)blockIdx.x * 4;
            o[0] = est0; o[1] = est1; o[2] = wall_clock64();
            unsigned long long* o2 = reinterpret_cast<unsigned long long*>(p.slabs) + 16384 + (size_t)blockIdx.x * 2;
            o2[0] = est2; o2[1] = est3;
        }
#endif
        return;
    }
    if (ep && vec) {
        // per 32-pixel row block: its TN*4 residual float4s first, then per accumulator quad vectors from LDS, arithmetic, stores
        // (all TM row blocks' residuals up front would not fit the register file of the large tiles)
        with_act(p.act, [&](auto AC) {
        constexpr int A = decltype(AC)::value;
#pragma unroll
        for (int i = 0; i < TL::TM; ++i) {
            const int m = m0 + (wm * TL::TM + i) * 32 + fr;
            if (m >= M) continue;
            const size_t row = (size_t)m * p.Cout;
            int cls = 0;
            size_t rrow = row;
            if (p.bias_cls || p.res_mode == (int)ResMode::UP2X) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                // BatchNorm shift folded in: taps that fall on the zero padding contribute none of it
                if (p.bias_cls) cls = 3 * (oy == 0 ? 0 : oy == p.Ho - 1 ? 2 : 1) + (ox == 0 ? 0 : ox == p.Wo - 1 ? 2 : 1);
                if (p.res_mode == (int)ResMode::UP2X) rrow = ((size_t)(n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout;
            }
            v4f r4[TL::TN][4];
            if (p.res_mode != (int)ResMode::NONE) {
#pragma unroll
                for (int j = 0; j < TL::TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = n0 + (wn * TL::TN + j) * 32 + 4 * fh2 + 8 * g;
                        r4[j][g] = co < p.Cout ? *reinterpret_cast<const v4f*>(res + rrow + co) : v4f{0.f, 0.f, 0.f, 0.f};
                    }
            }
#pragma unroll
            for (int j = 0; j < TL::TN; ++j) {
                const int cl = (wn * TL::TN + j) * 32 + 4 * fh2;           // channel within the tile
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = n0 + cl + 8 * g;
                    if (co >= p.Cout) continue;
                    const v4f b4 = *reinterpret_cast<const v4f*>(ep + cls * BN + cl + 8 * g);
                    v4f sl = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (A == (int)Act::PRELU) sl = *reinterpret_cast<const v4f*>(ep + 9 * BN + cl + 8 * g);
                    v4f v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = act_c<A>(acc[i][j][4 * g + c] + b4[c], sl[c]);
                    if (p.res_mode != (int)ResMode::NONE) v += r4[j][g];
                    if (out1) *reinterpret_cast<v4f*>(out1 + row + co) = v;
                    if (out2) {
                        const v4f s2 = *reinterpret_cast<const v4f*>(ep + 10 * BN + cl + 8 * g), t2 = *reinterpret_cast<const v4f*>(ep + 11 * BN + cl + 8 * g);
                        *reinterpret_cast<v4f*>(out2 + row + co) = v * s2 + t2;
                    }
                }
            }
        }
        });
        return;
    }
#pragma unroll
    for (int i = 0; i < TL::TM; ++i) {
        const int m = m0 + (wm * TL::TM + i) * 32 + fr;
        if (m >= M || (only_i >= 0 && i != only_i)) continue;
        const size_t row = (size_t)m * p.Cout;
        size_t rrow = row;
        const float* __restrict__ bias = p.bias;
        if (p.bias_cls) {                                   // BatchNorm shift folded in: taps that fall on the zero padding contribute none of it
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int cls = 3 * (oy == 0 ? 0 : oy == p.Ho - 1 ? 2 : 1) + (ox == 0 ? 0 : ox == p.Wo - 1 ? 2 : 1);
            bias += cls * p.Cout;
        }
        if (p.res_mode == (int)ResMode::UP2X) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            rrow = ((size_t)(n * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * p.Cout;
        }
#pragma unroll
        for (int j = 0; j < TL::TN; ++j) {
            if (only_j >= 0 && j != only_j) continue;
            const int cb = n0 + (wn * TL::TN + j) * 32 + 4 * fh2;
            if (vec) {
                v4f r4[4];
                if (p.res_mode != (int)ResMode::NONE) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = cb + 8 * g;
                        r4[g] = co < p.Cout ? *reinterpret_cast<const v4f*>(res + rrow + co) : v4f{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb + 8 * g;
                    if (co >= p.Cout) continue;
                    const v4f b4 = bias ? *reinterpret_cast<const v4f*>(bias + co) : v4f{0.f, 0.f, 0.f, 0.f};
                    v4f sl = {0.f, 0.f, 0.f, 0.f};
                    if (p.act == (int)Act::PRELU) sl = *reinterpret_cast<const v4f*>(p.slope + co);
                    v4f v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = apply_act(acc[i][j][4 * g + c] + b4[c], p.act, sl[c]);
                    if (p.res_mode != (int)ResMode::NONE) v += r4[g];
                    if (out1) *reinterpret_cast<v4f*>(out1 + row + co) = v;
                    if (out2) {
                        const v4f s2 = *reinterpret_cast<const v4f*>(p.s2 + co), t2 = *reinterpret_cast<const v4f*>(p.t2 + co);
                        *reinterpret_cast<v4f*>(out2 + row + co) = v * s2 + t2;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int co = cb + 8 * (e >> 2) + (e & 3);
                    if (co >= p.Cout) continue;
                    float v = apply_act(acc[i][j][e] + (bias ? bias[co] : 0.f), p.act, p.slope ? p.slope[co] : 0.f);
                    if (p.res_mode != (int)ResMode::NONE) v += res[rrow + co];
                    if (out1) out1[row + co] = v;
                    if (out2) out2[row + co] = v * p.s2[co] + p.t2[co];
                }
            }
        }
    }
}

// GRP: grouped form (several GEMMs stacked along M with one weight matrix each, see ConvArgs::wt_group_rows) — a separate
// instantiation, so that the ungrouped convolutions keep their register allocation.
// SC: with the folded-shortcut tap (ConvArgs::sc_in) — its own instantiation for the same reason.
template <int BM, int BN, int WM, int WN, int OCC, bool FAST, bool GRP, bool SC = false>
__global__ __launch_bounds__(WM * WN * 64, OCC) void conv_igemm_kernel(const ConvArgs p_by_value, const int tiles_n, const int chunks) {
    // The argument block is read where it already lies — the kernarg segment (constant address space, scalar loads) — instead of
    // through the by-value parameter: clang materialises that one as a private copy which the optimiser usually removes again, and when
    // it does not (it stopped doing so when this kernel grew the folded-shortcut tap) all 384 bytes live in scratch memory and every
    // field access becomes a scratch load: -25 % on every convolution.
    // (the instantiations without that tap keep the by-value form: there the copy is optimised away and the fields sit in SGPRs)
    const ConvArgs& p = SC ? *(const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr() : p_by_value;      // first explicit argument = offset 0
    using TL = Tile<BM, BN, WM, WN>;
    constexpr int TM = TL::TM, TN = TL::TN, RP = TL::RP, AL = TL::AL, BL = TL::BL;
    __shared__ v4f lds[2][(BM + BN) * 8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keep it scalar
    const int wm = wid / WN, wn = wid % WN;
    const int HoWo = p.Ho * p.Wo;
    const int M = p.B * HoWo;
    const int Ktot = p.ks * p.ks * p.Cin;
    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);      // source k-column of this lane (swizzle on the source side)
    const int fr = lane & 31, fh2 = lane >> 5;
    const int fsw = (fr >> 1) & 7;

    // ---- which tile(s) and which K range: plain tiles first, then helpers, then owners
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  Give every XCD
    // a CONTIGUOUS range of tiles instead, so that the tiles_n workgroups that read the same pixels (and the
    // neighbours that share their halo rows) hit the same L2 instead of fetching the input once per XCD.
    auto xcd_contiguous = [](int b, int n) {
        const int q = n >> 3, r = n & 7, x = b & 7;
        return x * q + min(x, r) + (b >> 3);
    };
    enum { PLAIN = 0, HELPER = 1, OWNER = 2 };
    int role = PLAIN, tile = 0, c_begin = 0, c_end = chunks;
    int hu = 0, hu_end = 0, helper_id = 0;                  // helper: its run of units [hu, hu_end)
    const int Kh = chunks - p.sk_owner_chunks;              // chunks per remainder tile that helpers compute
    if ((int)blockIdx.x < p.sk_full) {
        tile = xcd_contiguous(blockIdx.x, p.sk_full);
    } else if ((int)blockIdx.x < p.sk_full + p.sk_helpers) {
        role = HELPER;
        helper_id = xcd_contiguous(blockIdx.x - p.sk_full, p.sk_helpers);
        hu = helper_id * p.sk_q;
        hu_end = min(p.sk_units, hu + p.sk_q);
    } else {
        role = OWNER;
        tile = p.sk_full + xcd_contiguous(blockIdx.x - p.sk_full - p.sk_helpers, p.sk_rem);
        c_end = p.sk_owner_chunks;
    }
    unsigned* const sk_counters = reinterpret_cast<unsigned*>(p.slabs + SK_SLAB_FLOATS);
    // stream-K owner: "my hand-off wait timed out" flag.  The LAST word of the LDS images (free once the K loop is over, and beyond the
    // 12 * BN floats the epilogue parks there) — a variable of its own would cost the 256x64 / 128x32 tiles a workgroup per CU
    // (their images fill 80 / 40 KB exactly).
    volatile int* const sk_gave_up = reinterpret_cast<volatile int*>(&lds[1][0]) + ((BM + BN) * 8 * 4 - 1);

    do {
        int part = 0;
        if (role == HELPER) {                               // next segment of this helper's run
            const int r = hu / Kh, off = hu - r * Kh;
            const int len = min(Kh - off, hu_end - hu);
            tile = p.sk_full + r;
            c_begin = p.sk_owner_chunks + off; c_end = c_begin + len;
            part = helper_id - (r * Kh) / p.sk_q;           // position among the helpers that touch this tile
            hu += len;
        }
        const int gtile = tile + p.tile0;                   // (tile0 > 0: the first tiles of this convolution ran in conv_tall_kernel)
        const int tile_n = gtile % tiles_n, tile_m = gtile / tiles_n;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        // grouped form (Winograd's 36 GEMMs stacked along M, wt_group_rows rows each, a multiple of BM): only the weight
        // matrix depends on the group — inputs and outputs are one tall matrix
        const float* g_wt = p.wt;
        if (GRP && p.wt_group_rows > 0) g_wt += (size_t)(m0 / p.wt_group_rows) * p.wt_gs;

        // ---- loader bookkeeping.  LDS-DMA (global_load_lds_dwordx4) writes lane l of a wave at
        // wave-uniform base + 16*l, i.e. pass i of wave `wid` fills rows i*RP + wid*8 + (l>>3), 16-byte
        // column l&7 — the LDS image stays lane-linear and the XOR swizzle is applied to the SOURCE:
        // the lane fetches k-column (l&7) ^ ((row>>1)&7)  (cdna_hip_programming.md §5.4 rule 21).
        const float* a_ptr[AL];
        const float* sc_ptr[AL];                            // folded shortcut (ConvArgs::sc_in): this row's pixel of the 1x1 convolution's input
        unsigned a_mask[AL];
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int m = m0 + lrow + i * RP;
            a_ptr[i] = p.zeros; a_mask[i] = 0; sc_ptr[i] = p.zeros;
            if (p.ks == 1 && p.stride == 1) {               // plain GEMM rows (1x1 convs, FC, gallery): no pixel arithmetic at all
                if (m < M) { a_ptr[i] = p.in + (long)m * p.Cin + lqs * 4; a_mask[i] = 1u; }
            } else if (m < M) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                a_ptr[i] = p.in + (long)(((n * p.H + iy0) * p.W + ix0) * p.Cin) + lqs * 4;
                if (SC) sc_ptr[i] = p.sc_in + (long)(((n * p.sc_H + oy * p.sc_stride) * p.sc_W + ox * p.sc_stride) * p.sc_C) + lqs * 4;
                unsigned mk = 0;
                if (p.ks == 3) {
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        if ((unsigned)(iy0 + t / 3) < (unsigned)p.H && (unsigned)(ix0 + t % 3) < (unsigned)p.W) mk |= 1u << t;
                } else {
                    mk = 1u;
                }
                a_mask[i] = mk;
            }
        }
        // weights: wave-uniform base that advances 128 B per chunk + a constant 32-bit lane offset
        // (lets the loads use the scalar-base addressing form: no per-chunk vector pointer math)
        const char* w_base = reinterpret_cast<const char*>(g_wt) + ((size_t)n0 * p.Kpad + (size_t)c_begin * 32) * 4;
        unsigned w_off[BL];
#pragma unroll
        for (int i = 0; i < BL; ++i) w_off[i] = (unsigned)(((lrow + i * RP) * p.Kpad + lqs * 4) * 4);

        const unsigned long long zero_addr = (unsigned long long)p.zeros + (unsigned long long)(lane & 0) ;
        v4f* const dstA = &lds[0][wid * 64];               // + buf*(BM+BN)*8 + i*RP*8; the hardware adds 16*lane
        v4f* const dstB = &lds[0][BM * 8 + wid * 64];
        // FAST path (Cin % 32 == 0): a chunk never straddles a tap, so tap / channel position are
        // scalars that advance incrementally and the per-row source pointers are only rebuilt
        // (valid tap -> real address, padded tap -> zero line) when the tap changes.
        int ld_tap = 0, ld_ci = 0;
        const float* cur[AL];
        int adv[AL];                                        // floats a row's pointer advances per chunk: 32, or 0 while it reads the zero line
        // (loop-carried uses of the argument block go through locals: `p` lives in memory and every barrier / asm memory clobber in
        //  the K loop would otherwise re-fetch the field — an s_load + lgkmcnt wait per chunk)
        const int Cin = p.Cin, ksz = p.ks, inW = p.W;
        const int taps = ksz * ksz;
        auto tap_setup = [&]() __attribute__((always_inline)) {                            // (a dead row of the 25 088-deep FC would otherwise walk 100 KB past the 8 KiB of zeros)
            if (ld_tap >= taps) {                           // past the own taps: the folded shortcut's pixels (ConvArgs::sc_in)
                if (!SC) return;
#pragma unroll
                for (int i = 0; i < AL; ++i) {
                    const bool on = a_mask[i] != 0;
                    cur[i] = (const float*)(on ? (unsigned long long)(sc_ptr[i] + ld_ci) : zero_addr);
                    adv[i] = on ? 32 : 0;
                }
                return;
            }
            const int ky = ld_tap / 3, kx = ld_tap - ky * 3;
            const int toff = (ksz == 3 ? (ky * inW + kx) * Cin : 0) + ld_ci;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const unsigned long long real = (unsigned long long)(a_ptr[i] + toff);
                const bool on = ((a_mask[i] >> ld_tap) & 1u) != 0;
                cur[i] = (const float*)(on ? real : zero_addr);
                adv[i] = on ? 32 : 0;
            }
        };
        if (FAST) {
            ld_tap = min((c_begin * 32) / Cin, taps);
            ld_ci = c_begin * 32 - ld_tap * Cin;
            tap_setup();
        }
        auto load_chunk = [&](int kc, int buf) __attribute__((always_inline)) {
            v4f* const dA = dstA + buf * ((BM + BN) * 8);
            v4f* const dB = dstB + buf * ((BM + BN) * 8);
            if (FAST) {
#pragma unroll
                for (int i = 0; i < AL; ++i) { lds_dma16(cur[i], dA + i * RP * 8); cur[i] += adv[i]; }
#pragma unroll
                for (int i = 0; i < BL; ++i) lds_dma16(reinterpret_cast<const float*>(w_base + w_off[i]), dB + i * RP * 8);
                w_base += 128;
                ld_ci += 32;
                if (ld_ci >= Cin && ld_tap < taps) { ld_ci = 0; ++ld_tap; tap_setup(); }      // wave-uniform branch
                return;
            }
            const int kb = kc * 32;
            int tap, toff;                               // toff: float offset of this lane's 4 channels from a_ptr
            if (ksz == 1) { tap = 0; toff = kb; }
            else {
                const int k4 = kb + lqs * 4;
                tap = k4 / Cin;
                const int ci = k4 - tap * Cin;
                const int ky = tap / 3, kx = tap - ky * 3;
                toff = (ky * inW + kx) * Cin + ci - lqs * 4;
            }
            const bool kvalid = kb + lqs * 4 < Ktot;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                // integer select (v_cndmask), not a branch: border lanes read the zero line
                const unsigned long long real = (unsigned long long)(a_ptr[i] + toff);
                const unsigned long long sel = (kvalid && ((a_mask[i] >> tap) & 1u)) ? real : zero_addr;
                lds_dma16((const float*)sel, dA + i * RP * 8);
            }
#pragma unroll
            for (int i = 0; i < BL; ++i) lds_dma16(reinterpret_cast<const float*>(w_base + w_off[i]), dB + i * RP * 8);
            w_base += 128;
        };

        v16f acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // the epilogue's per-channel vectors (see conv_epilogue: ep), fetched now into a few registers and parked in LDS once the K loop
        // has released it — no global read is left for the epilogue to wait on
        constexpr int EPN = (12 * BN + WM * WN * 64 - 1) / (WM * WN * 64);
        const bool use_ep = role != HELPER && p.n_outs == 0 && (p.Cout & 3) == 0;
        float epv[EPN];
#pragma unroll
        for (int k = 0; k < EPN; ++k) {
            const int e = tid + k * (WM * WN * 64), a = e / BN, c = e - a * BN, co = n0 + c;
            float v = 0.f;
            if (use_ep && a < 12 && co < p.Cout) {
                if (a < 9) { if (p.bias && (a == 0 || p.bias_cls)) v = p.bias[a * p.Cout + co]; }
                else if (a == 9) { if (p.act == (int)Act::PRELU) v = p.slope[co]; }
                else if (p.out2) v = a == 10 ? p.s2[co] : p.t2[co];
            }
            epv[k] = v;
        }

        auto compute = [&](int buf) {
            const v4f* X = lds[buf] + (wm * TM * 32 + fr) * 8;
            const v4f* Wt = lds[buf] + BM * 8 + (wn * TN * 32 + fr) * 8;
            v4f x[2][TM], w[2][TN];
            {
                const int col = fh2 ^ fsw;
#pragma unroll
                for (int i = 0; i < TM; ++i) x[0][i] = X[i * 32 * 8 + col];
#pragma unroll
                for (int j = 0; j < TN; ++j) w[0][j] = Wt[j * 32 * 8 + col];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                if (s < 3) {
                    const int col = (2 * (s + 1) + fh2) ^ fsw;
#pragma unroll
                    for (int i = 0; i < TM; ++i) x[nxt][i] = X[i * 32 * 8 + col];
#pragma unroll
                    for (int j = 0; j < TN; ++j) w[nxt][j] = Wt[j * 32 * 8 + col];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][j][e], x[cur][i][e], acc[i][j], 0, 0, 0);
            }
        };

        if (c_begin < c_end) {
            load_chunk(c_begin, 0);
            __syncthreads();                              // (drains vmcnt: the DMA of chunk 0 has landed)
            for (int kc = c_begin, it = 0; kc < c_end; ++kc, ++it) {
                const int buf = it & 1;
                if (kc + 1 < c_end) load_chunk(kc + 1, buf ^ 1);
                compute(buf);
                __syncthreads();                          // all reads of `buf` done, DMA into buf^1 landed
            }
        }

        if (role == HELPER) {
            const int r = tile - p.sk_full;
            float* __restrict__ slab = p.slabs + ((size_t)r * p.sk_maxp + part) * TL::SLAB;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        v4f v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        *reinterpret_cast<v4f*>(slab + ((size_t)((i * TN + j) * 4 + g) * TL::T + tid) * 4) = v;
                    }
            if (p.sk_owner_chunks > 0) {
                // publish (cdna_hip_programming.md Guideline 16, counter form): every wave drains its stores,
                // barrier, ONE lane releases at agent scope, waits, then bumps the tile's counter
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0 && !p.sk_test_drop) {          // (sk_test_drop: the watchdog test's lost publication)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add(sk_counters + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            continue;
        }
        if (role == OWNER) {
            const int r = tile - p.sk_full;
            const int nparts = ((r + 1) * Kh - 1) / p.sk_q - (r * Kh) / p.sk_q + 1;
            if (tid == 0) {                                 // ONE lane polls relaxed, then ONE acquire
                // Bounded wait: forward progress rests on helpers (lower block indices) being dispatched before owners, which HIP
                // does not promise.  Past sk_timeout (default 2 s of the constant 100 MHz clock — five orders of magnitude beyond any
                // real wait) the owner gives up: it reports (tile, arrivals seen, arrivals expected) through its Net's host-visible
                // error record, leaves its tile unwritten and exits, so that the launch drains and the next call on that handle
                // returns FH_ERR_DEVICE instead of the process hanging with the GPU lease.
                const unsigned long long t0 = wall_clock64();
                unsigned seen, spins = 0;
                bool arrived = true;
                while ((seen = __hip_atomic_load(sk_counters + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != (unsigned)nparts) {
                    __builtin_amdgcn_s_sleep(4);
                    if ((++spins & 255u) == 0 && ((wall_clock64() - t0) >> 16) > (unsigned long long)p.sk_timeout) { arrived = false; break; }
                }
                if (arrived) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(sk_counters + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                } else if (p.sk_err) {
                    __hip_atomic_store(p.sk_err + 1, (unsigned)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(p.sk_err + 2, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(p.sk_err + 3, (unsigned)nparts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(p.sk_err + 0, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                *sk_gave_up = arrived ? 0 : 1;
            }
            __syncthreads();
            if (*sk_gave_up) return;                        // (workgroup-uniform; an owner runs exactly one tile)
            for (int q = 0; q < nparts; ++q) {              // K order: own chunks first, then the helpers' runs
                const float* slab = p.slabs + ((size_t)r * p.sk_maxp + q) * TL::SLAB;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const v4f v = *reinterpret_cast<const v4f*>(slab + ((size_t)((i * TN + j) * 4 + g) * TL::T + tid) * 4);
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[i][j][4 * g + c] += v[c];
                        }
            }
        }
        float* const ep = reinterpret_cast<float*>(&lds[0][0]);            // (every wave is past the K loop's last barrier: LDS is free)
        if (use_ep) {
#pragma unroll
            for (int k = 0; k < EPN; ++k) {
                const int e = tid + k * (WM * WN * 64);
                if (e < 12 * BN) ep[e] = epv[k];
            }
            __syncthreads();
        }
        // (the transposition scratch of the whole-line epilogue sits behind the 12 x BN vectors: both fit the tile buffers of every shape)
        static_assert((12 * BN + WM * WN * 32 * 36) * sizeof(float) <= sizeof(lds), "epilogue scratch");
        conv_epilogue<BM, BN, WM, WN>(p, acc, m0, n0, wm, wn, lane, -1, -1, use_ep ? ep : nullptr, use_ep && conv_ep_lines(p) ? ep + 12 * BN : nullptr);
    } while (role == HELPER && hu < hu_end);
}

// Owner-less form: sums the slabs of one remainder tile in K order and applies the epilogue.  One workgroup
// per (tile, 32x32 accumulator block of each wave): TM*TN times more workgroups than tiles, because this
// kernel is pure streaming and R < #CUs*2 tiles alone would leave most of the chip's load queues empty.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv_fixup_kernel(const ConvArgs p, const int tiles_n, const int chunks) {
    using TL = Tile<BM, BN, WM, WN>;
    constexpr int TM = TL::TM, TN = TL::TN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int r = blockIdx.x / (TM * TN);               // remainder tile index
    const int sub = blockIdx.x - r * (TM * TN);
    const int si = sub / TN, sj = sub - si * TN;
    const int tile = p.sk_full + r;
    const int nparts = ((r + 1) * chunks - 1) / p.sk_q - (r * chunks) / p.sk_q + 1;
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    v16f sum;
#pragma unroll
    for (int e = 0; e < 16; ++e) sum[e] = 0.f;
    for (int q = 0; q < nparts; ++q) {
        const float* __restrict__ slab = p.slabs + ((size_t)r * p.sk_maxp + q) * TL::SLAB;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4f v = *reinterpret_cast<const v4f*>(slab + ((size_t)((si * TN + sj) * 4 + g) * TL::T + tid) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) sum[4 * g + c] += v[c];
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            if (i == si && j == sj) acc[i][j] = sum;
    const int gtile = tile + p.tile0;                       // (see conv_igemm_kernel)
    const int tile_n = gtile % tiles_n, tile_m = gtile / tiles_n;
    conv_epilogue<BM, BN, WM, WN>(p, acc, tile_m * BM, tile_n * BN, wm, wn, lane, si, sj);
}

int conv_wt_rows(int Cout) { return (Cout + 127) / 128 * 128; }

// Packs plan-layout weights w[Cout][taps][Cin] into the kernel's [rows][Kpad] image, k = tap*Cin + ci.
// (A tap-inner order, k = (ci/32)*288 + tap*32 + ci%32, was measured: it turns the 9x input re-read
// into L1/L2 hits but needs the per-row source pointers rebuilt every chunk instead of every
// Cin/32 chunks, and those extra VALU instructions cost more than the cache hits gain: -3 %.)
void conv_pack_weights(const float* w, int Cout, int Cin, int ks, float* dst) {
    const int taps = ks * ks, Ktot = taps * Cin, Kpad = conv_kpad(Ktot);
    for (int co = 0; co < Cout; ++co)
        for (int k = 0; k < Ktot; ++k) dst[(size_t)co * Kpad + k] = w[(size_t)co * Ktot + k];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_tall_kernel (round 3) — 3x3 stride-1 pad-1 convolutions with Cin % 32 == 0 on maps up to 112 wide: ONE LDS image per 32-channel
// chunk serves all nine taps.
//
// conv_igemm_kernel gathers a fresh A tile per (tap, chunk): 256 rows x 128 B nine times per 32 channels, although the nine tiles are
// the same pixels shifted — for the row-linear tile [m0, m0 + BM) tap (ky, kx) reads pixel m + (ky - 1) W + (kx - 1).  What that costs is
// not bytes (they come from L2) but ISSUE: each 1 KB LDS-DMA piece takes its wave 60-185 cycles, 40 pieces per 64 MFMAs and wave for the
// 256x64 tile — the kernel that runs IResNet's stage 1 (Cin = 64: 2.2 ms of the 12.8 ms step) at 105-112 TFLOP/s where the 128x128
// tile of the wider layers reaches 117-123.  Here the A image holds the BM + 2 W + 2 consecutive pixels m0 - W - 1 .. m0 + BM + W of
// ONE 32-channel chunk (482 rows = 61.7 KB at W = 112, beside the 16 KB weight double buffer: still two workgroups per CU) and the nine
// taps are nine row-shifted views of it: fragment row = tile row + ky W + kx.  A traffic / 4.8, LDS-DMA pieces per MFMA / 2.6.
//   * K order: chunk-major (for each 32-channel chunk: nine taps); the weight chunk of (chunk c, tap t) sits at k = t Cin + 32 c of the
//     packed row, so only an address changes.
//   * a shifted view reads a REAL neighbour where the convolution pads with zero (left / right image border, first / last row, and the
//     pixels of the neighbouring image across a batch boundary): every lane carries a 9-bit validity mask of its fragment pixel per
//     32-row block and zeroes the fragment with v_cndmask (a select, not a multiply: the neighbour may hold anything).  Image rows
//     outside the tensor come from the zero line.
//   * swizzle: 16-byte column ^ ((LDS row >> 1) & 7), on the source side as in conv_igemm_kernel; a view's key is that of its shifted row.
//   * tiles are row-linear as in conv_igemm_kernel, so conv_epilogue (9 bias classes, PReLU, residual, second output) is shared and
//     nothing is wasted on maps that are no multiple of a spatial tile (56 = 3.5 x 16).
//   * no stream-K here: launch_cfg gives this kernel the whole ROUNDS of tiles and the remainder (tile0 = the first one) to
//     conv_igemm_kernel, whose owner / helper hand-off cuts those few tiles' K range over the whole chip.
template <int BM, int BN, int WM, int WN, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void conv_tall_kernel(const ConvArgs p, const int tiles_n, const int rows_a) {
    using TL = Tile<BM, BN, WM, WN>;
    constexpr int TM = TL::TM, TN = TL::TN, RP = TL::RP, BL = TL::BL;
    constexpr int NPMAX = (BM + 2 * 112 + 2 + RP - 1) / RP;                // loader passes of the tallest image (W = 112)
    extern __shared__ v4f tsm[];
    v4f* const As = tsm;                                                   // [rows_a][8 float4], rows_a = whole loader passes
    v4f* const Bs = tsm + (size_t)rows_a * 8;                              // [2][BN][8]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int W = p.W, HW = p.H * p.W, Cin = p.Cin;
    const int M = p.B * HW;
    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);                         // (RP % 16 == 0: the key of row i * RP + lrow is that of lrow)
    const int fr = lane & 31, fh2 = lane >> 5;
    int tile;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, x = blockIdx.x & 7;
        tile = x * q + min(x, r) + (int)(blockIdx.x >> 3);                 // XCD-contiguous tile order (as conv_igemm_kernel)
    }
    const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int np = rows_a / RP;

    // ---- A image loader: pass i fills LDS rows i * RP + lrow <- pixel m0 - W - 1 + row (all of one 32-channel chunk, this lane's column)
    const float* a_ptr[NPMAX];
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) {
        const long q = (long)m0 - W - 1 + i * RP + lrow;
        a_ptr[i] = (i < np && q >= 0 && q < M) ? p.in + q * Cin + lqs * 4 : p.zeros;
    }
    int a_adv[NPMAX];
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) a_adv[i] = a_ptr[i] == p.zeros ? 0 : 32;      // (rows that read the zero line do not walk off it)
    const char* w_tile = reinterpret_cast<const char*>(p.wt) + (size_t)n0 * p.Kpad * 4;
    unsigned w_off[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) w_off[i] = (unsigned)(((lrow + i * RP) * p.Kpad + lqs * 4) * 4);
    v4f* const dstA = As + wid * 64;
    v4f* const dstB = Bs + wid * 64;
    auto load_b = [&](int c, int tap, int buf) __attribute__((always_inline)) {
        const char* wb = w_tile + (size_t)(tap * Cin + c * 32) * 4;
#pragma unroll
        for (int i = 0; i < BL; ++i) lds_dma16(reinterpret_cast<const float*>(wb + w_off[i]), dstB + buf * (BN * 8) + i * RP * 8);
    };

    // ---- fragment views: block i of this wave = tile rows (wm TM + i) 32 + fr; validity of the nine taps of that pixel
    unsigned vmask[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 32 + fr;
        unsigned mk = 0;
        if (m < M) {
            const int rem = m % HW, oy = rem / W, ox = rem - oy * W;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if ((unsigned)(oy + t / 3 - 1) < (unsigned)p.H && (unsigned)(ox + t % 3 - 1) < (unsigned)W) mk |= 1u << t;
        }
        vmask[i] = mk;
    }
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // the epilogue's per-channel vectors (see conv_epilogue: ep), parked in LDS after the K loop
    constexpr int EPN = (12 * BN + WM * WN * 64 - 1) / (WM * WN * 64);
    float epv[EPN];
#pragma unroll
    for (int k = 0; k < EPN; ++k) {
        const int e = tid + k * (WM * WN * 64), a = e / BN, c = e - a * BN, co = n0 + c;
        float v = 0.f;
        if (a < 12 && co < p.Cout) {
            if (a < 9) { if (p.bias && (a == 0 || p.bias_cls)) v = p.bias[a * p.Cout + co]; }
            else if (a == 9) { if (p.act == (int)Act::PRELU) v = p.slope[co]; }
            else if (p.out2) v = a == 10 ? p.s2[co] : p.t2[co];
        }
        epv[k] = v;
    }

    const int NC = Cin >> 5;
    const int rbase = (wm * TM * 32 + fr);                                 // this lane's row in the tile (block 0)
    for (int c = 0; c < NC; ++c) {
        __syncthreads();                                                   // every wave is done with the previous chunk's image and weight buffers
        for (int i = 0; i < np; ++i) { lds_dma16(a_ptr[i < NPMAX ? i : 0], dstA + i * RP * 8); }
#pragma unroll
        for (int i = 0; i < NPMAX; ++i) a_ptr[i] += a_adv[i];
        load_b(c, 0, 0);
        __syncthreads();                                                   // (drains vmcnt: image + first weight chunk have landed)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int buf = tap & 1;
            if (tap + 1 < 9) load_b(c, tap + 1, buf ^ 1);
            const int shift = (tap / 3) * W + tap % 3;
            const int r0 = rbase + shift;                                  // LDS row of block 0's fragment pixel under this tap
            const int key = (r0 >> 1) & 7;                                 // (block i adds 32 i rows: same key)
            const v4f* X = As + r0 * 8;
            const v4f* Wt = Bs + buf * (BN * 8) + (wn * TN * 32 + fr) * 8;
            const int wkey = (fr >> 1) & 7;
            bool ok[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) ok[i] = (vmask[i] >> tap) & 1u;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                v4f x[TM], w[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    x[i] = X[i * 32 * 8 + ((2 * s2 + fh2) ^ key)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[i][e] = ok[i] ? x[i][e] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) w[j] = Wt[j * 32 * 8 + ((2 * s2 + fh2) ^ wkey)];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], x[i][e], acc[i][j], 0, 0, 0);
            }
            if (tap + 1 < 9) __syncthreads();                              // next weight chunk landed; everyone done with this one
        }
    }
    __syncthreads();                                                       // K loop over: LDS is free
    float* const ep = reinterpret_cast<float*>(Bs);
#pragma unroll
    for (int k = 0; k < EPN; ++k) {
        const int e = tid + k * (WM * WN * 64);
        if (e < 12 * BN) ep[e] = epv[k];
    }
    __syncthreads();
    conv_epilogue<BM, BN, WM, WN>(p, acc, m0, n0, wm, wn, lane, -1, -1, ep);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_pw_kernel (round 3) — 1x1 stride-1 convolutions as the plain GEMM they are: out [M][N] = in [M][K] * W^T.
// conv_igemm_kernel spends ~650 vector + ~490 scalar instructions per wave on tap / padding / index bookkeeping around the MFMAs of a
// short-K tile (that is why the Winograd GEMMs got wino_gemm_kernel); SCRFD's 1x1 convolutions (K = 72 / 152 / 288, 0.8 ms of the
// detector at B = 128) and MobileFaceNet's ran at 65-88 TFLOP/s in it.  Here: wino_gemm_kernel's loop — rows are contiguous, a lane's
// source address only advances by 32 floats per chunk — with a zero-line source for the float4 columns behind K in the last chunk
// (K % 4 == 0, any remainder), 128 x BN tiles with BN = 96 (N = 288: three exact column tiles) / 64 / 32, and the shared
// conv_epilogue (bias, ReLU / PReLU, residual incl. the FPN's 2x upsampled one, second output) with its vectors parked in LDS.
// M % 128 == 0 (whole tiles) and at least one tile per CU, else launch_conv keeps the generic kernel and its stream-K.
template <int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_pw_kernel(const ConvArgs p, const int tiles_n, const int chunks) {
    constexpr int BM = 128, TN = BN / 32, AL = BM / 32, BL = BN / 32;
    __shared__ v4f lds[2][(BM + BN) * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;
    const int nb = gridDim.x, q = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
    const int tile = x * q + min(x, r8) + (blockIdx.x >> 3);                // XCD-contiguous tile order
    const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = p.Cin;

    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);                          // source k-column of this lane (swizzle on the source side)
    const float* a_src = p.in + (size_t)(m0 + lrow) * K + lqs * 4;
    const float* b_src = p.wt + (size_t)(n0 + lrow) * p.Kpad + lqs * 4;
    const size_t a32 = (size_t)32 * K, b32 = (size_t)32 * p.Kpad;
    const float* const zsrc = p.zeros + lqs * 4;
    int kleft = K - lqs * 4;                                                // > 0: this lane's float4 of the current chunk lies inside a row
    v4f* const dstA = &lds[0][wid * 64];
    v4f* const dstB = &lds[0][BM * 8 + wid * 64];
    auto load_chunk = [&](int buf) {
        v4f* const dA = dstA + buf * ((BM + BN) * 8);
        v4f* const dB = dstB + buf * ((BM + BN) * 8);
        const bool in_k = kleft > 0;
#pragma unroll
        for (int i = 0; i < AL; ++i) lds_dma16(in_k ? a_src + i * a32 : zsrc, dA + i * 32 * 8);
#pragma unroll
        for (int i = 0; i < BL; ++i) lds_dma16(b_src + i * b32, dB + i * 32 * 8);
        a_src += 32; b_src += 32; kleft -= 32;
    };
    v16f acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;

    // the epilogue's per-channel vectors (see conv_epilogue: ep), parked in LDS after the K loop
    constexpr int EPN = (12 * BN + 255) / 256;
    float epv[EPN];
#pragma unroll
    for (int k = 0; k < EPN; ++k) {
        const int e = tid + k * 256, a = e / BN, c = e - a * BN, co = n0 + c;
        float v = 0.f;
        if (a < 12 && co < p.Cout) {
            if (a < 9) { if (p.bias && (a == 0 || p.bias_cls)) v = p.bias[a * p.Cout + co]; }
            else if (a == 9) { if (p.act == (int)Act::PRELU) v = p.slope[co]; }
            else if (p.out2) v = a == 10 ? p.s2[co] : p.t2[co];
        }
        epv[k] = v;
    }

#ifdef FACEHIP_PW_ABL
    // Diagnostic build (scripts/pw_ablate.sh, scripts/pw_phases.py; an ablated run's results are garbage).  p.sk_test_drop bits: 1 = no loads
    // in the K loop, 2 = no LDS reads, 4 = no barriers in the K loop, 8 = no epilogue, 16 = epilogue stores from one lane only,
    // 32 = the CU's second workgroup starts (bits 8..) x 0.64 us late, 64 = wait for the stores' acknowledgement before the last stamp.
    // Phase stamps (100 MHz ticks) of the 20x20x288 layers go to the stream-K workspace: [workgroup][entry, first chunk landed, K loop
    // done, epilogue done] and, from 8192 on, [workgroup][epilogue entry, first block in LDS, last store issued, ep vectors parked].
    const int abl = p.sk_test_drop;
    const unsigned long long ts0 = wall_clock64();
    if ((abl & 32) && blockIdx.x >= 256u && blockIdx.x < 512u)
        while (wall_clock64() - ts0 < (unsigned long long)(abl >> 8) * 64ull) __builtin_amdgcn_s_sleep(32);
    load_chunk(0);
    __syncthreads();
    const unsigned long long ts1 = wall_clock64();
    v4f xk = lds[0][(wid * 32 + fr) * 8 + (fh2 ^ fsw)], wk[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) wk[j] = lds[0][BM * 8 + fr * 8 + j * 32 * 8 + (fh2 ^ fsw)];
    for (int kc = 0; kc < chunks; ++kc) {
        const int buf = (abl & 1) ? 0 : (kc & 1);
        if (!(abl & 1) && kc + 1 < chunks) load_chunk(buf ^ 1);
        const v4f* X = lds[buf] + (wid * 32 + fr) * 8;
        const v4f* Wt = lds[buf] + BM * 8 + fr * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = (2 * s + fh2) ^ fsw;
            v4f xv = xk;
            v4f w[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = wk[j];
            if (!(abl & 2)) {
                xv = X[col];
#pragma unroll
                for (int j = 0; j < TN; ++j) w[j] = Wt[j * 32 * 8 + col];
            }
            asm volatile("" : "+v"(xv));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], xv[e], acc[0][j], 0, 0, 0);
        }
        if (!(abl & 4)) __syncthreads();
    }
    __syncthreads();
    const unsigned long long ts2 = wall_clock64();
    if ((abl & 8) && acc[0][0][0] != 123.456f) return;
#else
    load_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < chunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < chunks) load_chunk(buf ^ 1);
        const v4f* X = lds[buf] + (wid * 32 + fr) * 8;
        const v4f* Wt = lds[buf] + BM * 8 + fr * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = (2 * s + fh2) ^ fsw;
            const v4f xv = X[col];
            v4f w[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) w[j] = Wt[j * 32 * 8 + col];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], xv[e], acc[0][j], 0, 0, 0);
        }
        __syncthreads();
    }
#endif
    float* const ep = reinterpret_cast<float*>(&lds[0][0]);
#pragma unroll
    for (int k = 0; k < EPN; ++k) {
        const int e = tid + k * 256;
        if (e < 12 * BN) ep[e] = epv[k];
    }
    __syncthreads();
    static_assert((12 * BN + 4 * 32 * 36) * sizeof(float) <= sizeof(lds), "epilogue scratch");
#ifdef FACEHIP_PW_ABL
    const bool stamp = tid == 0 && p.slabs && p.Cout == 288 && p.Cin == 288;
    if (stamp) reinterpret_cast<unsigned long long*>(p.slabs)[8192 + (size_t)blockIdx.x * 4 + 3] = wall_clock64();
#endif
    conv_epilogue<BM, BN, 4, 1>(p, acc, m0, n0, wid, 0, lane, -1, -1, ep, conv_ep_lines(p) ? ep + 12 * BN : nullptr);
#ifdef FACEHIP_PW_ABL
    if (stamp) {
        const unsigned long long ts3 = wall_clock64();                  // stores issued
        if (abl & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and acknowledged
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.slabs) + (size_t)blockIdx.x * 5;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = (abl & 64) ? (unsigned long long)wall_clock64() : ts3; o[4] = 1;
    }
#endif
}

template <int BN, int OCC>
static void launch_pw_cfg(const ConvArgs& a, long M, hipStream_t s) {
    const int tiles_n = (a.Cout + BN - 1) / BN;
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL((conv_pw_kernel<BN, OCC>), dim3((unsigned)((M / 128) * tiles_n)), dim3(256), 0, s, a, tiles_n, a.Kpad / 32);
    timer.end(s, 11, a.t_flops, a.t_bytes);
}

static int ep_direct_env() {                         // FACEHIP_EP_DIRECT=1: the epilogues store straight from the accumulator registers (A / B timing)
    static const int v = [] { const char* e = getenv("FACEHIP_EP_DIRECT"); return e ? atoi(e) : 0; }();
    return v;
}
// true = launched
static bool launch_pw(ConvArgs a, hipStream_t s) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("FACEHIP_CONV_PW"); on = e ? atoi(e) : 1; }       // (0 = conv_igemm_kernel everywhere: A / B timing)
    const long M = (long)a.B * a.Ho * a.Wo;
    if (!on || a.no_pw || a.ks != 1 || a.stride != 1 || a.pad != 0 || a.wt_group_rows > 0 || a.sc_in || a.n_outs > 0 || (a.Cin & 3) || (a.Cout & 3) ||
        (M & 127) || a.H != a.Ho || a.W != a.Wo || a.Kpad % 32 || a.Kpad < a.Cin)
        return false;
    a.zeros = conv_zero_line();
    a.ep_direct = ep_direct_env();
#ifdef FACEHIP_PW_ABL
    { static const int abl = [] { const char* e = getenv("FACEHIP_PW_ABL"); return e ? atoi(e) : 0; }(); a.sk_test_drop = abl; }
#endif
    const int cus = a.cus > 0 ? a.cus : conv_num_cus();
    auto cols = [&](int bn) { return (a.Cout + bn - 1) / bn * bn; };
    // widest tile whose padded columns cost <= 7 %; the 32-wide one otherwise (the generic kernel pads the same way)
    int bn = 32;
    if (cols(64) * 100 <= a.Cout * 107) bn = 64;
    if (cols(96) * 100 <= a.Cout * 107 && cols(96) <= cols(64)) bn = 96;
    const long tiles = (M / 128) * (cols(bn) / bn);
    if (tiles < (long)cus) return false;                                    // fewer tiles than CUs: the generic kernel cuts K over the chip
    if (bn == 96) launch_pw_cfg<96, 2>(a, M, s);
    else if (bn == 64) launch_pw_cfg<64, 3>(a, M, s);
    else launch_pw_cfg<32, 4>(a, M, s);
    return true;
}

static int tall_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_CONV_TALL"); v = e ? atoi(e) : 1; }       // (0 = conv_igemm_kernel everywhere: A / B timing)
    return v;
}

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

static int g_num_cus = 0;
static const float* g_zeros = nullptr;
static int num_cus() {
    if (!g_num_cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_num_cus = n;
    }
    return g_num_cus;
}
int conv_num_cus() { return num_cus(); }
const float* conv_zero_line() {                      // 8 KiB of zeros: a padded tap streams up to Cin*4 bytes from it
    if (!g_zeros) {
        void* p = nullptr;
        if (hipMalloc(&p, 8192) == hipSuccess) { (void)hipMemset(p, 0, 8192); g_zeros = (const float*)p; }
    }
    return g_zeros;
}

size_t conv_slab_floats() { return SK_SLAB_FLOATS + SK_COUNTERS; }   // slabs + one counter word per remainder tile
void conv_workspace_init(float* ws) { (void)hipMemset(ws + SK_SLAB_FLOATS, 0, SK_COUNTERS * sizeof(unsigned)); }
void conv_workspace_reset_async(float* ws, hipStream_t s) { (void)hipMemsetAsync(ws + SK_SLAB_FLOATS, 0, SK_COUNTERS * sizeof(unsigned), s); }

// ---- stream-K watchdog: a host-mapped (pinned, device-visible) record an owner writes when its hand-off wait times out
static unsigned* g_sk_err = nullptr;
static std::atomic<unsigned> g_sk_generation{0};      // (handles may be driven from different host threads)
static unsigned g_sk_timeout = (unsigned)((2000ull * 100000ull) >> 16);      // 2 s in units of 2^16 ticks of the 100 MHz wall clock
static int g_sk_test_drop = 0;
unsigned* conv_error_words() {
    if (!g_sk_err) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) { memset(p, 0, 64); g_sk_err = (unsigned*)p; }
    }
    return g_sk_err;
}
// One record per Net (allocated with its workspace) so that a time-out is reported by a call on the handle whose launch was abandoned;
// launches without one (the single-kernel test entry points) share the process-wide record above.  Records are never freed while the
// device may still write them: a Net parks its record on a free list at destruction.
static std::vector<unsigned*> g_sk_free;
static std::mutex g_sk_mu;
unsigned* conv_error_record_new() {
    {
        std::lock_guard<std::mutex> lk(g_sk_mu);
        if (!g_sk_free.empty()) { unsigned* r = g_sk_free.back(); g_sk_free.pop_back(); memset(r, 0, 64); return r; }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
    memset(p, 0, 64);
    return (unsigned*)p;
}
void conv_error_record_release(unsigned* r) {
    if (!r) return;
    std::lock_guard<std::mutex> lk(g_sk_mu);
    g_sk_free.push_back(r);
}
bool conv_error_pending(const unsigned* rec) {
    const volatile unsigned* e = rec;
    return e && e[0];
}
bool conv_take_error(std::string& msg, unsigned* rec) {
    volatile unsigned* e = rec;
    if (!e || !e[0]) return false;
    char buf[256];
    snprintf(buf, sizeof buf, "HIP error: stream-K hand-off timed out (remainder tile %u saw %u of %u helper arrivals): the launch was "
             "abandoned, its outputs are incomplete", e[1], e[2], e[3]);
    msg = buf;
    e[0] = 0;
    g_sk_generation.fetch_add(1, std::memory_order_relaxed);                                   // every workspace's counters are suspect: owners re-zero them before their next run
    return true;
}
unsigned conv_error_generation() { return g_sk_generation.load(std::memory_order_relaxed); }
static std::atomic<unsigned> g_sk_debug_gen{0};
unsigned conv_debug_generation() { return g_sk_debug_gen.load(std::memory_order_relaxed); }
void conv_debug_streamk(int drop_publish, int timeout_ms) {
    g_sk_debug_gen.fetch_add(1, std::memory_order_relaxed);
    g_sk_test_drop = drop_publish;
    g_sk_timeout = (unsigned)(((unsigned long long)(timeout_ms > 0 ? timeout_ms : 2000) * 100000ull) >> 16);
}

static int sk_b_ratio() {                              // fix-up form only when a tile is cut into >= ratio/2 pieces (tuning hook)
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_SK_B_RATIO"); v = e ? atoi(e) : 4; }
    return v;
}
static int g_sk_min_owner = -1;
static int sk_min_owner() {
    if (g_sk_min_owner < 0) {
        const char* e = getenv("FACEHIP_SK_MIN_OWNER");   // tuning hook
        g_sk_min_owner = e ? atoi(e) : 12;
    }
    return g_sk_min_owner;
}
static int g_sk_margin2 = -1;                         // helpers' head start over the owners, in half chunks
static int sk_margin2() {
    if (g_sk_margin2 < 0) {
        const char* e = getenv("FACEHIP_SK_MARGIN");      // tuning hook
        g_sk_margin2 = e ? atoi(e) : 3;
    }
    return g_sk_margin2;
}

template <int BM, int BN, int WM, int WN, int OCC>
static void launch_cfg_tail(ConvArgs a, int resident_per_cu, int cfg_tag, hipStream_t s, const int T, const int tiles_n, const int S);

template <int BM, int BN, int WM, int WN, int OCC>
static void launch_cfg(ConvArgs a, int resident_per_cu, int cfg_tag, hipStream_t s) {
    const long M = (long)a.B * a.Ho * a.Wo;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (a.Cout + BN - 1) / BN;
    const bool grouped = a.wt_group_rows > 0;
    const int T = tiles_m * tiles_n;
    const int S = (a.cus > 0 ? a.cus : num_cus()) * resident_per_cu;
    a.zeros = conv_zero_line();
    a.tile0 = 0;
    // 3x3 stride-1 layers on maps up to 112 wide: the whole ROUNDS of tiles run in conv_tall_kernel (one LDS image per 32-channel chunk
    // for all nine taps), the remainder below with tile0 = the first tile it did not take
    // (256x64: IResNet's stage 1 / 2.  128x32: SCRFD's merged 64 -> 30 head convolutions at B = 128 — 340 -> 330 us on 80x80, 98 -> 91 on
    // 40x40: a small gain, because with three phase-locked workgroups per CU the image load at the top of each of the two chunks stays
    // exposed; a three-deep weight ring (two taps ahead, barrier with vmcnt(1)) changed nothing: 306 against 307 us)
    if constexpr ((BM == 256 && BN == 64) || (BM == 128 && BN == 32)) {
        constexpr int TOCC = BM == 256 ? 2 : 3;                             // resident workgroups per CU of the tall form (LDS)
        const int St = (a.cus > 0 ? a.cus : num_cus()) * TOCC;
        if (tall_enabled() && !grouped && !a.sc_in && a.ks == 3 && a.stride == 1 && a.pad == 1 && (a.Cin & 31) == 0 &&
            a.H == a.Ho && a.W == a.Wo && a.W <= 112 && (BN == 32 ? a.n_outs > 0 || (a.Cout & 3) == 0 : a.n_outs == 0 && (a.Cout & 3) == 0) &&
            a.res_mode != (int)ResMode::UP2X && T >= St) {
            constexpr int RPt = Tile<BM, BN, WM, WN>::RP;
            const int rows_a = (BM + 2 * a.W + 2 + RPt - 1) / RPt * RPt;
            const size_t lds = ((size_t)rows_a * 8 + 2 * BN * 8) * sizeof(v4f);
            int tall_tiles = (T / St) * St;
            tall_tiles -= tall_tiles % tiles_n;
            // A matrix-core-bound launch takes as long as its busiest CU has tiles (measured on the Winograd GEMMs), not whole
            // chip-wide rounds: when the remainder is at most one more tile per CU and a tile is short (< ~30 us of one CU's matrix cores),
            // running it here costs less than the second launch + fix-up of the split (SCRFD's 80x80 head convolution: 6 400 tiles = 25 per
            // CU exactly: 329 + 38 us -> 343 us); IResNet's 256x64 tiles (44 us each) keep the split.
            {
                const int cus_ = a.cus > 0 ? a.cus : num_cus();
                const double tile_us = 2.0 * BM * BN * (double)a.Kpad / 0.43e6;      // ~110 TFLOP/s over 256 CUs
                if (tall_tiles > 0 && T - tall_tiles <= cus_ && tile_us < 30.0) tall_tiles = T;
            }
            if (lds <= (size_t)(160 / TOCC) * 1024 && tall_tiles > 0) {
                static bool attr_set = false;
                if (!attr_set) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tall_kernel<BM, BN, WM, WN, TOCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
                    attr_set = true;
                }
                KernelTimer& tt = KernelTimer::get();
                tt.begin(s);
                hipLaunchKernelGGL((conv_tall_kernel<BM, BN, WM, WN, TOCC>), dim3((unsigned)tall_tiles), dim3(WM * WN * 64), lds, s, a, tiles_n, rows_a);
                tt.end(s, 10, a.t_flops * ((double)tall_tiles / T), a.t_bytes * ((double)tall_tiles / T));
                a.t_flops *= (double)(T - tall_tiles) / T; a.t_bytes *= (double)(T - tall_tiles) / T;
                a.tile0 = tall_tiles;
                if (tall_tiles == T) return;
            }
        }
    }
    return launch_cfg_tail<BM, BN, WM, WN, OCC>(a, resident_per_cu, cfg_tag, s, T - a.tile0, tiles_n, S);
}

template <int BM, int BN, int WM, int WN, int OCC>
static void launch_cfg_tail(ConvArgs a, int resident_per_cu, int cfg_tag, hipStream_t s, const int T, const int tiles_n, const int S) {
    const bool grouped = a.wt_group_rows > 0;
    const int chunks = a.Kpad / 32;
    int full = (T / S) * S;
    int R = T - full;
    int helpers = 0, owners = 0;
    bool fixup = false;
    a.sk_units = 0; a.sk_q = 1; a.sk_owner_chunks = 0; a.sk_maxp = 1;
    // (a matrix-core-bound launch ends with its busiest CU: leaving R remainder tiles whole costs (ceil(R / CUs) - R / CUs) tile times of
    //  imbalance; cutting them over the chip costs a fix-up launch (10-17 us).  Short tiles — SCRFD's 20x20 head convolutions, 11 us —
    //  stay whole: 31 + 17 us -> 25 us; IResNet's 44 us tiles keep the split.)
    const int cus_ = a.cus > 0 ? a.cus : num_cus();
    const double tile_us = 2.0 * BM * BN * (double)a.Kpad / 0.43e6;                  // ~110 TFLOP/s over 256 CUs
    const bool cheap_whole = ((R + cus_ - 1) / cus_ - (double)R / cus_) * tile_us < 10.0;
    if (R > 0 && R * 10 <= S * 9 && a.slabs && a.sk_enable && !cheap_whole) {   // worth it only if the last round is <= 90 % full
        // (1) owners + helpers inside the launch
        const int H = S - R;
        const int q_o = (int)(((long)R * chunks + (long)H * sk_margin2() / 2 + S - 1) / S);
        const int Kh = chunks - q_o;
        // (the hand-off costs an owner ~8 us — fence, poll, slab reads: only worth it next to >= 12 chunks of its own work)
        if (Kh >= 1 && q_o >= sk_min_owner() && R <= (int)SK_COUNTERS) {
            const long U = (long)R * Kh;
            const int q_h = (int)((U + H - 1) / H);
            const int maxp = (Kh + q_h - 1) / q_h + 1;
            if (maxp <= 3 && (size_t)R * maxp * BM * BN <= SK_SLAB_FLOATS) {
                a.sk_units = (int)U; a.sk_q = q_h; a.sk_owner_chunks = q_o; a.sk_maxp = maxp;
                helpers = (int)((U + q_h - 1) / q_h); owners = R;
            }
        }
        // (2) otherwise: every segment to a slab, reduced by the fix-up kernel
        if (!owners) {
            const long U = (long)R * chunks;
            int q = (int)((U + S - 1) / S);
            // do not cut segments shorter than 8 chunks — 24 for very deep K (the 25 088-deep FC: 61 slabs per tile made the fix-up kernel,
            // 16 workgroups summing them one after the other, take 35 of the layer's 80 us; 21 slabs: 48 + 20 us)
            // (the tail of a convolution whose whole rounds ran in conv_tall_kernel is a few tiles on an otherwise empty chip: cut them finer)
            static int tail_q = -1;
            if (tail_q < 0) { const char* e = getenv("FACEHIP_TALL_TAIL_Q"); tail_q = e ? atoi(e) : 3; }
            const int min_q = a.tile0 > 0 ? std::min(chunks, tail_q) : chunks < 8 ? chunks : chunks >= 256 ? 24 : 8;
            if (q < min_q) q = min_q;
            const int maxp = (chunks + q - 1) / q + 1;
            if (q * sk_b_ratio() <= chunks * 2 && q < chunks && (size_t)R * maxp * BM * BN <= SK_SLAB_FLOATS) {
                a.sk_units = (int)U; a.sk_q = q; a.sk_maxp = maxp;
                helpers = (int)((U + q - 1) / q); fixup = true;
            }
        }
    }
    if (!helpers) { full = T; R = 0; }
    a.sk_full = full; a.sk_helpers = helpers; a.sk_rem = R;
    a.sk_err = owners ? (a.sk_err ? a.sk_err : conv_error_words()) : nullptr; a.sk_timeout = g_sk_timeout; a.sk_test_drop = g_sk_test_drop;
    a.ep_direct = ep_direct_env();
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    const dim3 grid((unsigned)(full + helpers + owners));
    if (grouped) {                                             // (wt_group_rows % BM == 0 and Cin % 32 == 0: the caller's contract)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, OCC, true, true>), grid, dim3(WM * WN * 64), 0, s, a, tiles_n, chunks);
    } else if (a.sc_in)                                        // (launch_conv checked Cin % 32 == 0)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, OCC, true, false, true>), grid, dim3(WM * WN * 64), 0, s, a, tiles_n, chunks);
    else if ((a.Cin & 31) == 0)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, OCC, true, false>), grid, dim3(WM * WN * 64), 0, s, a, tiles_n, chunks);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, OCC, false, false>), grid, dim3(WM * WN * 64), 0, s, a, tiles_n, chunks);
    timer.end(s, grouped ? 7 : cfg_tag, a.t_flops, a.t_bytes);   // (grouped = Winograd GEMM: its own tag)
    if (fixup) {
        timer.begin(s);
        hipLaunchKernelGGL((conv_fixup_kernel<BM, BN, WM, WN>), dim3((unsigned)(R * (BM / WM / 32) * (BN / WN / 32))), dim3(WM * WN * 64), 0, s, a, tiles_n, chunks);
        timer.end(s, 6, 0.0, 0.0);
    }
}

void launch_conv(const ConvArgs& a, int cfg, hipStream_t s) {
    const long M = (long)a.B * a.Ho * a.Wo;
    if (M <= 0) return;
    // the loader and the epilogue index pixels and tensor elements with 32-bit integers
    if ((long)a.B * a.H * a.W * a.Cin >= (1L << 31) || M * a.Cout >= (1L << 31))
        throw std::runtime_error("conv: tensor too large for the 32-bit element offsets of one launch (split the batch)");
    if (a.sc_in && (a.ks != 3 || a.Cin % 32 != 0 || a.sc_C % 32 != 0 || a.Kpad != 9 * a.Cin + a.sc_C || a.wt_group_rows > 0 ||
                    (long)a.B * a.sc_H * a.sc_W * a.sc_C >= (1L << 31)))
        throw std::runtime_error("conv: folded shortcut needs a 3x3 convolution with Cin % 32 == 0, sc_C % 32 == 0 and Kpad = 9*Cin + sc_C");
    if (launch_pw(a, s)) return;
    if (cfg < 0) cfg = conv_pick_cfg(M, a.Cout);
    switch (cfg) {
        case 0: launch_cfg<128, 128, 2, 2, 2>(a, 2, 0, s); break;
        case 1: launch_cfg<256, 64, 4, 1, 2>(a, 2, 1, s); break;
        case 2: launch_cfg<128, 32, 4, 1, 4>(a, 4, 2, s); break;
        default: launch_cfg<64, 64, 2, 2, 4>(a, 5, 3, s); break;
    }
}

}  // namespace fh
