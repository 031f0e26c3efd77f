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

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <type_traits>

#include "kernels.h"
#include "plan.h"

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
template <int BN, int WM, int WN, int DS, int HC, bool DIRECT>
__global__ __launch_bounds__(256, DIRECT ? (HC == 4 ? 3 : BN * DS <= 64 ? 3 : 2) : HC == 4 ? 4 : 2) void dwpw_kernel(const ConvArgs p, const int tiles_x, const int tiles_y, const int tiles_n) {
    static_assert(DS == 1 || DIRECT, "stride 2 needs the direct form");
    constexpr int BM = DP_BM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int HALO_SLOTS = !DIRECT ? (DP_HALO * HC + 255) / 256 * 256 : 0;  // float4 slots, whole DMA passes
    constexpr int PPP = 256 / HC;                                          // pixels per depthwise pass
    constexpr int SR = BM / PPP;                                           // DIRECT: output rows per thread strip
    constexpr int NR = (SR - 1) * DS + 3;                                  //         input rows a strip touches
    __shared__ v4f lds[HALO_SLOTS + BM * 8 + BN * 8];
    __shared__ float bias_s[BN];                                           // pointwise bias of this tile's columns (see the epilogue)
    v4f* const halo = lds;
    // LDS-halo form: the spare slots behind the 180 halo pixels (the DMA passes cover whole multiples of 256) carry the chunk's 9
    // depthwise tap vectors + bias, [10][HC] float4: same DMA instructions, no extra LDS, and ten global loads per thread less
    // (2 560 per workgroup and chunk — more vector-memory instructions than the halo itself)
    const v4f* const dww_s = halo + DP_HALO * HC;
    static_assert(DIRECT || DP_HALO * HC + 10 * HC <= HALO_SLOTS, "no room for the depthwise weights behind the halo");
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
    const float* img = p.in + (size_t)n * p.H * p.W * C;  // H x W = depthwise input, Ho x Wo = output grid

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
    const int wslot_raw = (HP - 1) * 256 + tid - DP_HALO * HC;              // >= 0: slot behind the halo pixels
    const int wslot = (!DIRECT && wslot_raw >= 0 && wslot_raw < 10 * HC) ? wslot_raw : -1;
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

    // fetched now, parked in LDS under the first chunk's barrier: the epilogue then issues no global load between its stores (loads and
    // stores share the in-order vmcnt: a bias load after a store waits for that store's acknowledgement — 12 round trips per tile)
    const float bias_v = (tid < BN && n0 + tid < p.Cout) ? p.bias[n0 + tid] : 0.f;
    for (int kc = 0; kc < chunks; ++kc) {
        const int c0 = kc * 32;
        const bool hvalid = c0 + hq * 4 < C;                               // 16-byte columns past C are never read: no DMA for them
        const int ksteps = min(4, (C - c0 + 7) >> 3);                      // 8-deep MFMA steps that hold real channels
        __syncthreads();                                                   // previous chunk: halo + fragments fully consumed
        if (!DIRECT) {
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                const float* src = h_off[j] >= 0 ? img + h_off[j] + c0 : p.zeros;
                bool go = hvalid;
                if (j == HP - 1 && wslot >= 0) {                            // this lane's slot of the last pass holds a weight vector
                    const int k = wslot / HC, ch = c0 + (wslot - k * HC) * 4;
                    src = ch < C ? (k < 9 ? p.dw_w + (size_t)k * C + ch : p.dw_b + ch) : p.zeros;
                    go = true;
                }
                if (go) dwpw_dma16(src, halo + j * 256 + wid * 64);
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) dwpw_dma16(reinterpret_cast<const float*>(w_base + w_off[i]), Wt + i * 256 + wid * 64);
        w_base += 128;
        // depthwise weights / bias of this thread's 4 channels (channels >= C: zero line -> zero output)
        const int cw = c0 + dq * 4;
        const bool cvalid = cw < C;
        v4f wk[9], b4;
        if (DIRECT) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = *reinterpret_cast<const v4f*>(cvalid ? p.dw_w + (size_t)k * C + cw : p.zeros);
            b4 = *reinterpret_cast<const v4f*>(cvalid ? p.dw_b + cw : p.zeros);
        }
        if (!DIRECT) {
            __syncthreads();                                               // halo + weights landed (barrier drains vmcnt)
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[k] = dww_s[k * HC + dq];
            b4 = dww_s[9 * HC + dq];
            if (dq < 2 * ksteps) {
                // a lane owns NPX vertically adjacent pixels of one column: their 3x3 windows share rows, (NPX + 2) * 3 tap reads instead
                // of NPX * 9 (18 instead of 36 at 32 channels per chunk) — this phase is bound by LDS bandwidth
                constexpr int NPX = BM / PPP;
                const int pxx = dp & (DP_TW - 1), py0 = (dp / DP_TW) * NPX;
                v4f a[NPX];
#pragma unroll
                for (int i = 0; i < NPX; ++i) a[i] = b4;
                if (cvalid) {                                              // (the invalid half of a partly valid step: zeros, its halo was not loaded)
#pragma unroll
                    for (int r = 0; r < NPX + 2; ++r)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const v4f h = halo[((py0 + r) * DP_HW + pxx + kx) * HC + dq];
#pragma unroll
                            for (int i = 0; i < NPX; ++i)
                                if (r - i >= 0 && r - i < 3) a[i] += h * wk[(r - i) * 3 + kx];
                        }
                    if (p.dw_act == (int)Act::RELU) {
#pragma unroll
                        for (int i = 0; i < NPX; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[i][e] = a[i][e] > 0.f ? a[i][e] : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < NPX; ++i) {
                    const int px = (py0 + i) * DP_TW + pxx;
                    At[px * 8 + (dq ^ ((px >> 1) & 7))] = a[i];
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
        if (kc == 0 && tid < BN) bias_s[tid] = bias_v;
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
                    const v4f bb = *reinterpret_cast<const v4f*>(bias_s + (co - n0));
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
    if (a.dw_stride == 2 && a.Cin <= 16 && BN <= 64)     // 16 channels = 4 float4 columns: every depthwise lane busy, 2-row strips (fewer registers)
        hipLaunchKernelGGL((dwpw_kernel<(BN <= 64 ? BN : 64), (BN <= 64 ? WM : 2), (BN <= 64 ? WN : 2), 2, 4, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
    else if (a.dw_stride == 2) hipLaunchKernelGGL((dwpw_kernel<BN, WM, WN, 2, 8, true>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
    else if (a.Cin <= 16 && BN <= 64) hipLaunchKernelGGL((dwpw_kernel<(BN <= 64 ? BN : 64), (BN <= 64 ? WM : 2), (BN <= 64 ? WN : 2), 1, 4, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
    else hipLaunchKernelGGL((dwpw_kernel<BN, WM, WN, 1, 8, false>), grid, dim3(256), 0, s, a, tiles_x, tiles_y, tiles_n);
}

// a.in = depthwise input [B,H,W,C] (a.dw_stride 1 | 2, pad 1; Ho x Wo = its output grid = the pointwise grid), a.Cin = C, a.wt = packed pointwise weights
// [..][conv_kpad(C)], a.dw_w [9][C], a.dw_b [C]; activation of the pointwise part limited to NONE / RELU.

// ---------------------------------------------------------------------------------------------------------------------------------
// SCRFD's opening block in ONE persistent kernel: u8 frame -> stem 3x3 (3 -> 16 channels, stride 1 | 2, preprocess folded into the
// weights) -> depthwise 3x3 -> pointwise 1x1 (16 -> Cout <= 32).  The stem's 16-channel map — the largest tensor of the network — is
// never written.  A tile (8 x 16 outputs) is a chain of four short phases; run as one workgroup per tile that chain is latency, not
// bandwidth (measured: 6.8 us per tile, 4 workgroups per CU, 0.68 ms for 128 frames while moving 1 GB).  So the workgroups are
// persistent: weights, biases and MFMA fragments are fetched once, and the only global read of a tile — its u8 window — is issued one
// tile AHEAD into registers, so that no phase of the loop waits on memory:
//   1. the prefetched u8 window (9S+3 rows x (17S+3) pixels, as aligned dwords) goes registers -> LDS; the next tile's loads are issued
//   2. stem on the matrix cores, v_mfma_f32_16x16x32_bf16: rows = 16 output channels (A = weights), columns = 16 halo pixels (B),
//      K = the 27 taps of the 3 x 9-byte window (+5 zeros).  u8 values (and the 127.5 of the conv padding) are exact in bf16, every
//      fp32 weight travels as three bf16 terms (hi + mid + lo = w exactly: stem_pack_wfrag), every product is exact in fp32 and the
//      accumulator is fp32: the fp32 FMA chain's value up to summation order, at a tenth of its vector instructions.
//      Lane l of a group: pixel l & 15, K block g = l >> 4 (g < 3: bytes 0..7 of window row g; g = 3: byte 8 of rows 0..2); the result
//      comes back as pixel l & 15, channels 4g..4g+3 = one 16-byte column of the halo image [180 pixels][16 channels].
//   3. depthwise 3x3 from the halo image (vector ALU, 4 channels per lane) -> A tile [128 pixels][32 k], XOR-swizzled
//   4. pointwise product on v_mfma_f32_32x32x2_f32 (K = 16), bias + ReLU, stores.
// Three barriers per tile.  Tiles are dealt so that the workgroups of one XCD work on neighbouring tiles at the same time (shared
// window rows hit that XCD's L2).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Folded stem weights wf [27][16] (tap-major: (ky*9 + kx*3 + byte) x channel) as the A fragments of v_mfma_f32_16x16x32_bf16: three
// bf16 terms (hi, mid, lo — their sum is the fp32 weight exactly), lane l = channel l & 15, K block l >> 4; K index 8g + j = byte j of
// window row g (g < 3), 24 + r = byte 8 of row r, 27..31 = 0.  One [3][64][4]-dword block per 16 output channels.
void stem_pack_wfrag(const float* wf, int Cout, unsigned* out) {             // Cout % 16 == 0; out: [Cout / 16][3][64][4] dwords
    auto bf = [](float x) {                                                // round to nearest even, as bits
        unsigned u; memcpy(&u, &x, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    };
    auto fl = [](unsigned b) { unsigned u = b << 16; float x; memcpy(&x, &u, 4); return x; };
    for (int cb = 0; cb < Cout / 16; ++cb)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int co = cb * 16 + (l & 15), k = 8 * (l >> 4) + j;
                float w = 0.f;
                if (k < 24) w = wf[((k >> 3) * 9 + (k & 7)) * Cout + co];
                else if (k < 27) w = wf[((k - 24) * 9 + 8) * Cout + co];
                const unsigned hi = bf(w); const float r1 = w - fl(hi);
                const unsigned mid = bf(r1); const float r2 = r1 - fl(mid);
                const unsigned t[3] = {hi, mid, bf(r2)};
                for (int q = 0; q < 3; ++q) {
                    unsigned& d = out[((cb * 3 + q) * 64 + l) * 4 + (j >> 1)];
                    d = (j & 1) ? (d | (t[q] << 16)) : t[q];
                }
            }
}

constexpr int FR_PITCH = 32;                                               // dwords per staged u8 row: (17*2+3)*3 + 3 bytes <= 128
constexpr int FR_ROWS = 21;                                                // 9*2 + 3
constexpr int FR_SLOTS = (FR_ROWS * FR_PITCH + 255) / 256;                 // staged dwords per thread (3)

// Wave priority by tile count, for the persistent kernels below.  A SIMD's instruction arbiter prefers its OLDEST wave: with several
// persistent workgroups per CU and a static share of tiles each, the first-launched workgroup runs at full speed and the last-launched one
// on what is left (front_kernel's phase stamps: a workgroup's loop took 586 k / 690 k / 811 k / 936 k cycles by launch order on its CU) —
// the kernel ends with the slowest while the fast ones' slots idle.  s_setprio beats age, so every workgroup takes each level in turn.
__device__ __forceinline__ void rotate_wave_priority(int it) {
    switch (it & 3) {                                                      // (s_setprio takes an immediate)
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, which would make every barrier of the tile loop
// wait for the NEXT tile's window loads
__device__ __forceinline__ void front_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// STEP4: the frames' row pitch is a multiple of 4 — the misalignment of a staged row is then the same for every row of a tile (one
// scalar) instead of a per-row value.
template <bool STEP4>
__global__ __launch_bounds__(256, 4) void front_kernel(const ConvArgs p, const int tiles_x, const int tiles_y, const int tiles_total) {
    __shared__ unsigned stage[2][FR_ROWS * FR_PITCH];
    __shared__ v4f halo[192 * 5];                                          // [180 pixels][16 channels], pixel pitch 5 float4 (odd: lane = pixel reads spread over all banks)
    __shared__ v4f Wt[32 * 8];
    __shared__ v4f cst[48];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.u8_stride, step = p.u8_step;
    const int nrows = 9 * S + 3;
    const int row_bytes = p.u8_srcW * 3;

    // ---- once per workgroup: pointwise weights -> LDS, depthwise weights / biases -> LDS, stem fragments -> registers
    {
        const int lrow = tid >> 3, lqs = (tid & 7) ^ ((lrow >> 1) & 7);
        dwpw_dma16(p.wt + (size_t)lrow * p.Kpad + lqs * 4, Wt + wid * 64);
    }
    // depthwise weights [9][16], depthwise bias [16], pointwise bias [32] live in LDS (re-read per tile) — in registers they would push
    // the kernel past the 128 VGPRs that 4 workgroups per CU allow
    if (tid < 36) cst[tid] = *reinterpret_cast<const v4f*>(p.dw_w + tid * 4);
    else if (tid < 40) cst[tid] = *reinterpret_cast<const v4f*>(p.dw_b + (tid - 36) * 4);
    else if (tid < 48) cst[tid] = 4 * (tid - 40) < p.Cout ? *reinterpret_cast<const v4f*>(p.bias + 4 * (tid - 40)) : v4f{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, lp = lane & 15;                               // stem: K block / pixel of the group
    const v4u* wfrag = reinterpret_cast<const v4u*>(p.stem_wfrag);
    const v4u wq_hi = wfrag[lane], wq_mid = wfrag[64 + lane], wq_lo = wfrag[128 + lane];
    const v4f sbias = *reinterpret_cast<const v4f*>(p.stem_bf + 4 * g);
#if defined(__HIP_DEVICE_COMPILE__)
    // (an empty asm that READS them: the compiler places its own wait for these loads here, in the prologue — it cannot see the explicit
    //  s_waitcnt below and would otherwise wait at their first use, inside the tile loop, draining the window prefetch with them)
    asm volatile("" ::"v"(wq_hi), "v"(wq_mid), "v"(wq_lo), "v"(sbias));
#endif
    const bf16x8 w_hi = __builtin_bit_cast(bf16x8, wq_hi), w_mid = __builtin_bit_cast(bf16x8, wq_mid), w_lo = __builtin_bit_cast(bf16x8, wq_lo);
    const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;

    // ---- everything about a lane's three stem groups that does not depend on the tile: halo pixel (hy, hx), its slot in the staged
    // image (row rr, byte cb relative to the tile's window origin) and in the halo image
    constexpr int NG = ((DP_HALO + 15) / 16 + 3) / 4;                       // groups per wave (3)
    int g_hy[NG], g_hx[NG], g_rq[NG], g_cb[NG], g_hp[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int hp = (wid + 4 * i) * 16 + lp;
        g_hp[i] = hp < DP_HALO ? hp : -1;
        g_hy[i] = hp / DP_HW; g_hx[i] = hp - g_hy[i] * DP_HW;
        const int rr = g_hy[i] * S + (g < 3 ? g : 0);                      // g < 3: window row g;  g = 3: rows 0..2, byte 8
        g_rq[i] = rr;
        g_cb[i] = g_hx[i] * S * 3 + (g < 3 ? 0 : 8);
    }
    // a halo pixel lies inside the map and its 3x3 window inside the net input iff  AY0 <= ay <= AY1  and  AX0 <= ax <= AX1 (tiles whose
    // whole halo does take the plain path; the others patch the conv padding in, see the stem below)
    const int AY0 = 1, AY1 = min(p.H - 1, (p.u8_inH - 2) / S), AX0 = 1, AX1 = min(p.W - 1, (p.u8_inW - 2) / S);

    // ---- tiles of this workgroup: XCD x owns a contiguous run, its workgroups walk it side by side (t, t + wgs, ...); the tile
    // coordinates move by the same three steps every time: no division in the loop
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

    unsigned pf[FR_SLOTS];
    const int pr = tid / FR_PITCH, pd4 = (tid % FR_PITCH) * 4;             // this thread's slots: rows pr + 8k, dword pd
    auto prefetch = [&]() __attribute__((always_inline)) {                // the u8 window of tile (n, tyi, txi) -> registers (issued, not waited for)
        const uint8_t* frame = p.u8_src + (size_t)n * p.u8_img_stride;
        const int sy0 = (tyi * DP_TH - 1) * S - 1, sx3 = ((txi * DP_TW - 1) * S - 1) * 3;
        const unsigned base_lo = (unsigned)(unsigned long long)frame + (unsigned)(sy0 * step + sx3);     // low bits of the window origin's address
#pragma unroll
        for (int k = 0; k < FR_SLOTS; ++k) {
            const int r = pr + (256 / FR_PITCH) * k;
            const unsigned shr = (STEP4 ? base_lo : base_lo + (unsigned)(r * step)) & 3u;
            const int rel = sx3 - (int)shr + pd4;                            // byte position in the frame row
            // rows / dwords outside the pasted image read as 0 (the letterbox canvas).  A dword that straddles an end of the row is read
            // from the clamped position — the last / first four bytes of the row, an unaligned but in-bounds access — and shifted into
            // place when it is staged (stage_fix below), so every byte of the image is exact in the staged window
            const bool ok = r < nrows && (unsigned)(sy0 + r) < (unsigned)p.u8_srcH && rel > -4 && rel < row_bytes;
            const int relc = min(max(rel, 0), row_bytes - 4);
            pf[k] = ok ? *(const dwpw_gmem_u32*)(frame + (unsigned)((sy0 + r) * step + relc)) : 0u;
        }
    };

    // the prologue's loads and the weight DMA have landed — the only full memory wait of the kernel: the barriers of the tile loop
    // order LDS traffic only (front_barrier), so a tile's prefetch stays in flight across them
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    front_barrier();
    if (t < run1) prefetch();
    int buf = 0;
#ifdef FACEHIP_DWPW_PROF
    long long fph[6] = {0, 0, 0, 0, 0, 0}; long long fst0; int fnt = 0;
    const long long fclk0 = __builtin_readcyclecounter(), frt0 = __builtin_amdgcn_s_memrealtime();   // shader clock vs the constant 100 MHz counter
#define FRONT_STAMP(i) { const long long now_ = __builtin_readcyclecounter(); fph[i] += now_ - fst0; fst0 = now_; }
#else
#define FRONT_STAMP(i)
#endif
    int prio_it = (int)(blockIdx.x >> 8);                                  // (launch order on the CU, as far as the dispatcher deals round-robin)
    for (; t < run1; t += wgs, buf ^= 1) {
        if (!p.no_prio) rotate_wave_priority(prio_it++);                   // (see rotate_wave_priority; no_prio: A / B switch)
#ifdef FACEHIP_DWPW_PROF
        fst0 = __builtin_readcyclecounter(); ++fnt;
#endif
        const int cn = n, cty0 = tyi * DP_TH, ctx0 = txi * DP_TW;          // this tile (n / tyi / txi move on to the prefetched one)
        const uint8_t* frame = p.u8_src + (size_t)cn * p.u8_img_stride;
        const int sy0 = (cty0 - 1) * S - 1, sx0 = (ctx0 - 1) * S - 1;
        const unsigned base_lo = (unsigned)(unsigned long long)frame + (unsigned)(sy0 * step + sx0 * 3);
        // 1. window: registers -> LDS (waits for the loads issued a whole tile ago), then the next tile's loads
        // (the whole halo in the map and every window inside the net input: AY1 <= H - 1, AX1 <= W - 1)
        const bool tile_inside = cty0 - 1 >= AY0 && cty0 + DP_TH <= AY1 && ctx0 - 1 >= AX0 && ctx0 + DP_TW <= AX1;
#pragma unroll
        for (int k = 0; k < FR_SLOTS; ++k) {
            unsigned v = pf[k];
            if (sx0 * 3 < 4 || sx0 * 3 + 4 * FR_PITCH > row_bytes) {        // (wave-uniform: the window reaches an end of the image's rows — the
                                                                            //  frame's border or the letterbox edge) clamped dwords: bytes into place
                const int r = pr + (256 / FR_PITCH) * k;
                const unsigned shr = (STEP4 ? base_lo : base_lo + (unsigned)(r * step)) & 3u;
                const int rel = sx0 * 3 - (int)shr + pd4;
                const int dl = min(max(rel, 0), row_bytes - 4) - rel;       // > 0: straddles the row's start, < 0: its end
                const unsigned sh = 8u * (unsigned)min(abs(dl), 3);
                v = dl > 0 ? v << sh : dl < 0 ? v >> sh : v;
            }
            if (tid + 256 * k < FR_ROWS * FR_PITCH) stage[buf][tid + 256 * k] = v;
        }
        FRONT_STAMP(0)
        advance();
        if (t + wgs < run1) prefetch();
        FRONT_STAMP(1)
        front_barrier();
        FRONT_STAMP(2)
        // 2. stem, every pixel from the staged image — no global access in this loop, so nothing here waits on the prefetch (a global load
        // anywhere in it would make the compiler drain vmcnt at its join: the NEXT tile's window, in every group).  Tiles that touch the
        // border of the map (wave-uniform test) patch the convolution's zero padding in: a byte OUTSIDE the net input counts as 127.5 —
        // it cancels against the folded (v - 127.5) / 128 — while the letterbox canvas (inside the net input, outside the pasted image)
        // is the u8 0 the staged window already holds.  Round 2 / early round 3 sent those pixels through a second, per-byte pass over
        // global memory: border tiles (14.5 % of a 640x640 frame's) cost twice an inner tile and the slowest wave ran 20 % above the mean.
        const unsigned* st = stage[buf];
        auto stem_finish = [&](const float (&f)[8], bool inmap, bool store, int hp) __attribute__((always_inline)) {
            v4u bq;                                                         // fp32 -> bf16 by truncation: exact for 0..255 and 127.5
#pragma unroll
            for (int i = 0; i < 4; ++i)
                bq[i] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, f[2 * i + 1]), __builtin_bit_cast(unsigned, f[2 * i]), 0x07060302u);
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, bq);
            v4f a4 = sbias;
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_lo, bfrag, a4, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_mid, bfrag, a4, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w_hi, bfrag, a4, 0, 0, 0);
            if (p.stem_act == (int)Act::RELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a4[c] = a4[c] > 0.f ? a4[c] : 0.f;
            }
            if (!inmap) a4 = v4f{0.f, 0.f, 0.f, 0.f};                       // outside the map: the depthwise zero padding
            if (store) halo[hp * 5 + g] = a4;
        };
        static_assert(4 * (NG - 1) * 16 + 3 * 16 < DP_HALO, "every wave has NG groups: the loop has no early exit (straight-line code: the three groups interleave)");
#pragma unroll
        for (int i = 0; i < NG; ++i) {                                      // (wave-uniform trip count: the MFMAs run with all lanes)
            const int hp = g_hp[i];
            bool inmap = hp >= 0;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = 0.f;
            if constexpr (STEP4) {
                // One instruction sequence for all four K blocks (the two-path form below costs a wave BOTH paths: ~100 instructions per
                // group, a third of them exec-mask bookkeeping): three dword reads — consecutive dwords of window row g (g < 3), or the
                // dword holding byte 8 in each of the three rows (g = 3: lane-constant stride `kstep`) — and three byte permutes with
                // per-lane selectors: g < 3: alignbyte twice (+ an identity), g = 3: byte sh of d0 / d1 / d2 into bytes 0 / 1 / 2.  The
                // g = 3 lanes' bytes 3..7 are whatever the dwords hold — finite u8 values against the ZERO weights of K = 27..31.  Lanes
                // beyond the halo (hp < 0) read the window's first dwords; their result is never stored.
                const unsigned pos = (base_lo & 3u) + (unsigned)(hp >= 0 ? g_cb[i] : 0);
                const unsigned sh = pos & 3u;
                const int i0 = (hp >= 0 ? g_rq[i] * FR_PITCH : 0) + (int)(pos >> 2);
                const int kstep = g < 3 ? 1 : FR_PITCH;
                const unsigned d0 = st[i0], d1 = st[i0 + kstep], d2 = st[i0 + 2 * kstep];
                const unsigned alsel = 0x03020100u + 0x01010101u * sh;
                const unsigned selA = g < 3 ? alsel : (sh | ((4u + sh) << 8));
                const unsigned selC = g < 3 ? 0x03020100u : (0x03000100u | ((4u + sh) << 16));
                unsigned n0 = __builtin_amdgcn_perm(d1, d0, selA);
                const unsigned n1 = __builtin_amdgcn_perm(d2, d1, alsel);
                n0 = __builtin_amdgcn_perm(d2, n0, selC);
                f[0] = (float)(n0 & 255u); f[1] = (float)((n0 >> 8) & 255u); f[2] = (float)((n0 >> 16) & 255u); f[3] = (float)(n0 >> 24);
                f[4] = (float)(n1 & 255u); f[5] = (float)((n1 >> 8) & 255u); f[6] = (float)((n1 >> 16) & 255u); f[7] = (float)(n1 >> 24);
            } else if (hp >= 0) {
                if (g < 3) {                                                // 8 bytes at an unaligned position: 3 aligned dwords
                    const unsigned pos = ((STEP4 ? base_lo : base_lo + (unsigned)(g_rq[i] * step)) & 3u) + (unsigned)g_cb[i];
                    const unsigned* q = st + g_rq[i] * FR_PITCH + (pos >> 2);
                    const unsigned sh = pos & 3u;
                    const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
                    const unsigned n0 = __builtin_amdgcn_alignbyte(d1, d0, sh), n1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    f[0] = (float)(n0 & 255u); f[1] = (float)((n0 >> 8) & 255u); f[2] = (float)((n0 >> 16) & 255u); f[3] = (float)(n0 >> 24);
                    f[4] = (float)(n1 & 255u); f[5] = (float)((n1 >> 8) & 255u); f[6] = (float)((n1 >> 16) & 255u); f[7] = (float)(n1 >> 24);
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const unsigned pos = ((STEP4 ? base_lo : base_lo + (unsigned)((g_rq[i] + r) * step)) & 3u) + (unsigned)g_cb[i];
                        f[r] = (float)((st[(g_rq[i] + r) * FR_PITCH + (pos >> 2)] >> (8u * (pos & 3u))) & 255u);
                    }
                }
            }
            if (!tile_inside) {
                const int ay = cty0 + g_hy[i] - 1, ax = ctx0 + g_hx[i] - 1;
                inmap = inmap && (unsigned)ay < (unsigned)p.H && (unsigned)ax < (unsigned)p.W;
                const int iyb = ay * S - 1, ixb = ax * S - 1;               // window origin in the net input
                const bool c0 = (unsigned)ixb >= (unsigned)p.u8_inW, c1 = (unsigned)(ixb + 1) >= (unsigned)p.u8_inW, c2 = (unsigned)(ixb + 2) >= (unsigned)p.u8_inW;
                if (g < 3) {                                                // bytes 0..7 of window row g: columns 0 0 0 1 1 1 2 2
                    const bool ro = (unsigned)(iyb + g) >= (unsigned)p.u8_inH;
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = (ro || (j < 3 ? c0 : j < 6 ? c1 : c2)) ? 127.5f : f[j];
                } else {                                                    // byte 8 (column 2) of rows 0..2
#pragma unroll
                    for (int r = 0; r < 3; ++r) f[r] = ((unsigned)(iyb + r) >= (unsigned)p.u8_inH || c2) ? 127.5f : f[r];
                }
            }
            stem_finish(f, inmap, hp >= 0, hp);
        }
        FRONT_STAMP(3)
        front_barrier();
        // 3 + 4 merged (round 3, as dwpw_reg_kernel): a wave owns 32 pixels, lane = (pixel fr, half fh2); the depthwise 3x3 of channels
        // 8 j + 4 fh2 .. + 3 of ITS pixel is the B fragment of the pointwise MFMAs — no A tile, no barrier between the two, and the
        // 16 KB the A tile took are gone from LDS.  Accumulators start from the pointwise bias.  One MFMA per slot, the next
        // step's tap reads RA tap-slots ahead (ring registers), pinned by sched_barrier.
        {
            const int pr_ = wid * 32 + fr, py = pr_ / DP_TW, pxx = (py & 1) ? (pr_ - py * DP_TW - 2) & (DP_TW - 1) : pr_ - py * DP_TW;   // (rotation: see dwpw_reg_kernel)
            const v4f* const hb = halo + (py * DP_HW + pxx) * 5 + fh2;
            const v4f* const db = cst + fh2;                                // [tap][4] + 2 j ; bias at tap 9
            v16f acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f pb = cst[40 + fh2 + 2 * q];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[4 * q + c] = pb[c];
            }
            constexpr int NT = 10, RA = 2, RING = 3, STEPS = 2;            // (a deeper ring spills: the kernel lives within 128 VGPRs for 4 workgroups per CU)
            v4f hv[RING], dv[RING];
            auto issue = [&](int G) __attribute__((always_inline)) {
                const int jj = G / NT, tt = G % NT;
                if (jj >= STEPS) return;
                dv[G % RING] = db[(tt == 0 ? 9 : tt - 1) * 4 + 2 * jj];
                if (tt > 0) hv[G % RING] = hb[(((tt - 1) / 3) * DP_HW + (tt - 1) % 3) * 5 + 2 * jj];
            };
            v4f an;
            auto consume = [&](int G) __attribute__((always_inline)) {
                if (G / NT >= STEPS) return;
                if (G % NT == 0) an = dv[G % RING]; else an += hv[G % RING] * dv[G % RING];
            };
            const float dfl = p.dw_act == (int)Act::RELU ? 0.f : -INFINITY, ofl = p.act == (int)Act::RELU ? 0.f : -INFINITY;
            const v4f* Wp = Wt + fr * 8;
#pragma unroll
            for (int G = 0; G < RA; ++G) issue(G);
#pragma unroll
            for (int G = 0; G < NT; ++G) { issue(G + RA); consume(G); }
#pragma unroll
            for (int j = 0; j < STEPS; ++j) {
                v4f a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = fmaxf(an[e], dfl);
                const v4f w = Wp[(2 * j + fh2) ^ fsw];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[sl], a[sl], acc, 0, 0, 0);
#pragma unroll
                    for (int G = NT * (j + 1) + sl * NT / 4; G < NT * (j + 1) + (sl + 1) * NT / 4; ++G) { issue(G + RA); consume(G); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            FRONT_STAMP(4)
            const int oy = cty0 + py, ox = ctx0 + pxx;
            if (oy < p.Ho && ox < p.Wo) {
                float* __restrict__ orow = p.out1 + (((size_t)cn * p.Ho + oy) * p.Wo + ox) * p.Cout;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = 4 * fh2 + 8 * q;
                    if (co >= p.Cout) continue;
                    v4f v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[4 * q + c], ofl);
                    *reinterpret_cast<v4f*>(orow + co) = v;
                }
            }
            FRONT_STAMP(5)
        }
    }
#ifdef FACEHIP_DWPW_PROF
    if (lane == 0 && p.slabs) {
        long long* o = reinterpret_cast<long long*>(p.slabs) + ((size_t)blockIdx.x * 4 + wid) * 8;
        for (int i = 0; i < 6; ++i) o[i] = fph[i];
        o[6] = fnt;
        o[7] = (__builtin_readcyclecounter() - fclk0) * 1000 / ((long long)__builtin_amdgcn_s_memrealtime() - frt0 + 1);   // shader cycles per 100 MHz tick x 1000 = MHz x 10
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Register-fed form of the stride-1 block (round 3): depthwise 3x3 -> pointwise 1x1 with NO A tile in LDS and NO barrier in the K loop.
//
// dwpw_kernel above spends its time in its phase chain, not on bandwidth or arithmetic (phase timers, 80x80x72: DMA issue 11 k,
// depthwise 8.9 k, prologue 9.5 k cycles per tile): per 32-channel chunk every wave waits at two barriers, the depthwise result makes a
// round trip through LDS, a chunk with few live channels (40 = 32 + 8) idles three quarters of the depthwise lanes, and every tile pays
// the prologue again.  Here:
//   * a wave owns 32 pixels of the 8 x 16 tile for the WHOLE K loop: lane = (pixel r = lane & 31, half h = lane >> 5).  For the 8-channel
//     step j it evaluates the depthwise 3x3 (+bias +ReLU) of channels 8j + 4h .. + 3 of ITS pixel straight from the halo image — a
//     float4 that is exactly the B fragment v_mfma_f32_32x32x2_f32 wants from that lane (k = 8j + e for h = 0, 8j + 4 + e for h = 1,
//     e = 0..3): the result goes from the vector ALU into the matrix core without touching LDS.  Waves never wait for each other
//     inside a tile, so one wave's depthwise FMAs / LDS reads overlap another's MFMAs on the same SIMD.
//   * the halo holds ALL channels of the tile at once ([180 pixels][C + 4 floats]: pitch 4 * odd, so the 16-byte reads of
//     neighbouring pixels spread over all banks), the pointwise weights sit in LDS in fragment order for the kernel's lifetime,
//     depthwise taps + biases too.
//   * persistent workgroups; the NEXT tile's halo is fetched global -> registers while this tile computes (front_kernel's scheme: no
//     second halo buffer, so the occupancy stays), written to LDS between two LDS-only barriers.
//   * the halo comes in through BUFFER loads with a per-image descriptor: rows above / below the image are out of range and read as
//     zero in hardware, columns left / right of it are pushed out of range by one select (edge tiles only) — no per-item bounds
//     arithmetic, no branches, and the loads of a tile issue back to back.
//   * accumulators start from the pointwise bias; the epilogue is ReLU + float4 stores.
// Phase stamps (scripts/dwpw_prof.sh, 80x80x72 -> 72, cycles per tile and workgroup): barriers 1.4 k, registers -> LDS 0.8 k, prefetch
// issue 1.8 k, K loop 14.7 k (= two waves per SIMD sharing the matrix pipe at 6.9 k of MFMA issue each: the loop is MFMA-bound; a
// third of those MFMAs multiply the padding 72 -> 96 columns), stores 2.4 k.
// Measured and dropped: a straight-line epilogue (template-constant store count, lanes outside the map storing to a sink) so that the
// compiler's wait for the prefetched registers becomes the exact `vmcnt(stores + later loads)` instead of vmcnt(0): the ISA showed
// vmcnt(21)..(9) as intended, the layers ran no faster and the 16-channel layers slower.
// C % 8 == 0 (CQ = C / 4 even), Cout % 4 == 0, Cout <= 32 * TN.
template <int CQ, int TN, int OCC, int DS>
__global__ __launch_bounds__(256, OCC) void dwpw_reg_kernel(const ConvArgs p, const int tiles_x, const int tiles_y, const int tiles_total) {
    static_assert(CQ % 2 == 0, "whole 8-channel MFMA steps");
    constexpr int C = CQ * 4, STEPS = C / 8;
    constexpr int PQ = CQ + 1;                                             // halo pixel pitch in float4 (odd)
    // halo of an 8 x 16 output tile under a depthwise stride DS: rows 2 oy - 1 .. 2 oy + 1 -> (8 - 1) DS + 3 rows, same for the columns
    constexpr int HH = (DP_TH - 1) * DS + 3, HWD = (DP_TW - 1) * DS + 3, HALO = HH * HWD;   // 10 x 18 = 180 (DS = 1), 17 x 33 = 561 (DS = 2)
    constexpr int NPF = (HALO * CQ + 255) / 256;                           // prefetched float4 per thread
    // LDS slot of halo pixel hp = hy * HWD + hx.  Stride 1: hp itself.  Stride 2 (round 5): a row's EVEN columns first, then its odd ones —
    // lane = output pixel reads input column 2 px + kx, i.e. slot px + (kx >> 1) of plane kx & 1: neighbouring lanes are ONE pixel pitch
    // (odd in float4) apart.  With interleaved columns they were two apart: every halo index of an instruction had the same parity, 8 bank
    // quads for the 16 lanes of a ds_read_b128 group, a 2-way conflict on every tap (SQ_LDS_BANK_CONFLICT 0.34 of the LDS cycles, round 4).
    constexpr int HW2 = (HWD + 1) / 2;
    auto slot = [](int hp) { if (DS == 1) return hp; const int hy = hp / HWD, hx = hp - hy * HWD; return hy * HWD + (hx & 1) * HW2 + (hx >> 1); };
    auto tapoff = [](int ky, int kx) { return DS == 1 ? ky * HWD + kx : ky * HWD + (kx & 1) * HW2 + (kx >> 1); };
    extern __shared__ v4f smem[];
    v4f* const halo = smem;                                                // [HALO][PQ]
    v4f* const dwl = halo + HALO * PQ;                                  // [10][CQ]: 9 taps + bias
    v4f* const Wl = dwl + 10 * CQ;                                         // [STEPS][2][Cout]: A fragments (n = row, 4 k of half h)
    float* const pwb = reinterpret_cast<float*>(Wl + STEPS * 2 * p.Cout);  // [32 * TN] pointwise bias (zero behind Cout)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int Cout = p.Cout;
    // ReLU (floor 0) or none (floor -inf) as one v_max: a branch per step would end the basic block and with it the overlap of the steps
    const float dw_floor = p.dw_act == (int)Act::RELU ? 0.f : -INFINITY, out_floor = p.act == (int)Act::RELU ? 0.f : -INFINITY;

    // ---- once per workgroup
    for (int i = tid; i < 10 * CQ; i += 256) {
        const int k = i / CQ, q = i - k * CQ;
        dwl[i] = *reinterpret_cast<const v4f*>(k < 9 ? p.dw_w + (size_t)k * C + 4 * q : p.dw_b + 4 * q);
    }
    for (int i = tid; i < STEPS * 2 * Cout; i += 256) {
        const int n = i % Cout, jh = i / Cout;                             // jh = 2 j + h
        Wl[i] = *reinterpret_cast<const v4f*>(p.wt + (size_t)n * p.Kpad + 4 * jh);
    }
    for (int i = tid; i < 32 * TN; i += 256) pwb[i] = i < Cout ? p.bias[i] : 0.f;

    // this lane's pixel and fragment addresses (float4 units)
    // (stride 1: the wave's second pixel row takes its columns rotated by 2.  A ds_read_b128 serves lanes {0-3, 12-15, 20-27} and
    //  {4-11, 16-19, 28-31} together; with an odd pixel pitch the 16 addresses fall into 16 different bank quads iff the pixels' linear halo
    //  indices differ mod 16, and a halo row is 18 = 16 + 2 pixels: unrotated, lanes 12 / 13 collide with lanes 26 / 27 and lanes 4 / 5 with
    //  lanes 18 / 19 — every fragment read took 8 LDS cycles instead of 4 (SQ_LDS_BANK_CONFLICT = 27-34 % of SQ_LDS_IDX_ACTIVE, round-4 counters))
    //  (stride 2 with its column planes: consecutive slots again and a row pitch of 2 * 33 = 66 = 64 + 2 slots — the same rotation)
    const int pix = wid * 32 + r, py = pix / DP_TW, px = (py & 1) ? (pix - py * DP_TW - 2) & (DP_TW - 1) : pix - py * DP_TW;
    const v4f* const hbase = halo + (py * DS * HWD + px) * PQ + h;            // + tapoff(ky, kx) * PQ + 2 j   (stride 2: px indexes a column plane)
    const v4f* const dbase = dwl + h;                                      // + tap * CQ + 2 j
    int wrow[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) wrow[jn] = h * Cout + min(32 * jn + r, Cout - 1);   // rows >= Cout: any valid address (their columns are never stored)

    // ---- tiles: XCD x owns a contiguous run, its workgroups walk it side by side
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, wgs = gridDim.x >> 3;       // (gridDim.x is a multiple of 8)
    const int q8 = tiles_total >> 3, r8 = tiles_total & 7;
    const int run0 = xcd * q8 + min(xcd, r8), run1 = run0 + q8 + (xcd < r8 ? 1 : 0);
    const int per_img = tiles_x * tiles_y;

    // halo of tile t: global -> registers (issued, not waited for).  Per thread and item the byte offset relative to the tile's halo
    // origin is tile-invariant (voff); per tile: one add each, and on tiles that touch the left / right border a select.
    v4f pf[NPF];
    int voff[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int i = min(tid + 256 * k, HALO * CQ - 1);
        const int hp = i / CQ, q = i - hp * CQ;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        voff[k] = ((hy * p.W + hx) * C + 4 * q) * 4;
    }
    const int img_bytes = p.H * p.W * C * 4;
    auto prefetch = [&](int t) __attribute__((always_inline)) {
        const int n = t / per_img, rem = t - n * per_img;
        const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
        const int y0 = tyi * DP_TH * DS - 1, x0 = txi * DP_TW * DS - 1;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + (size_t)n * p.H * p.W * C), 0, img_bytes, 0x00020000);
        const int tile_off = (y0 * p.W + x0) * C * 4;                       // (negative on the first tile row / column: wraps out of range)
        int vo[NPF];
#pragma unroll
        for (int k = 0; k < NPF; ++k) vo[k] = tile_off + voff[k];
        if (x0 < 0 || x0 + HWD > p.W) {                                   // wave-uniform: only the first / last tile column
#pragma unroll
            for (int k = 0; k < NPF; ++k) {
                const int i = min(tid + 256 * k, HALO * CQ - 1);
                const int hx = (i / CQ) % HWD;
                vo[k] = (unsigned)(x0 + hx) < (unsigned)p.W ? vo[k] : (int)0x80000000;
            }
        }
#pragma unroll
        for (int k = 0; k < NPF; ++k) pf[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo[k], 0, 0));
    };
    int t = run0 + wg;
    if (t < run1) prefetch(t);
    front_barrier();                                                        // dwl / Wl / pwb written by all waves
#ifdef FACEHIP_DWPW_PROF
    long long ph[6] = {0, 0, 0, 0, 0, 0}; long long st0; int ntiles = 0;
    const long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#define DWPW_STAMP(i) { const long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - st0; st0 = now_; }
#else
#define DWPW_STAMP(i)
#endif
    int prio_it = (int)(blockIdx.x >> 8);
    for (; t < run1; t += wgs) {
        if (!p.no_prio) rotate_wave_priority(prio_it++);                   // (oldest-first arbitration vs static tile shares: see rotate_wave_priority)
        const int n = t / per_img, rem = t - n * per_img;
        const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
        const int ty0 = tyi * DP_TH, tx0 = txi * DP_TW;
#ifdef FACEHIP_DWPW_PROF
        st0 = __builtin_readcyclecounter(); ++ntiles;
#endif
        front_barrier();                                                    // every wave is done reading the previous tile's halo
        DWPW_STAMP(0)
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = tid + 256 * k;
            if (i < HALO * CQ) { const int hp = i / CQ; halo[slot(hp) * PQ + (i - hp * CQ)] = pf[k]; }   // pixel pitch PQ = CQ + 1
        }
        DWPW_STAMP(1)
        if (t + wgs < run1) prefetch(t + wgs);
        DWPW_STAMP(2)
        front_barrier();                                                    // halo complete
        DWPW_STAMP(3)

        v16f acc[TN];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f b = *reinterpret_cast<const v4f*>(pwb + 32 * jn + 8 * g + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[jn][4 * g + c] = b[c];
            }
        // ---- K loop, hand-scheduled.  Left to itself the compiler emits, per step, nine times (2 ds_read, wait, 2 FMA) and only then the
        // step's MFMAs (ds_read, wait, 4 MFMA): every LDS latency exposed, nothing beside the matrix pipe.  So the order is written
        // out: one MFMA per SLOT; behind it the slot's share of the NEXT step's depthwise — each tap's two LDS reads are issued RA
        // tap-slots ahead of their FMAs (ring registers hv / dv), the next 32-column group's weight fragment four slots ahead — and a
        // sched_barrier(0) pins the slot.  The counted lgkmcnt waits the compiler inserts are then exact (LDS returns in order).
        constexpr int NS = 4 * TN;                                          // MFMA slots per 8-channel step
        constexpr int NT = 10;                                              // tap-slots per step: the depthwise bias, then the 9 taps
        constexpr int RA = 4, RING = 5;
        v4f hv[RING], dv[RING], wq[2];
        auto issue = [&](int G) __attribute__((always_inline)) {           // LDS reads of global tap-slot G = NT * step + t
            const int jj = G / NT, tt = G % NT;
            if (jj >= STEPS) return;
            dv[G % RING] = dbase[(tt == 0 ? 9 : tt - 1) * CQ + 2 * jj];
            if (tt > 0) hv[G % RING] = hbase[tapoff((tt - 1) / 3, (tt - 1) % 3) * PQ + 2 * jj];
        };
        v4f an;
        auto consume = [&](int G) __attribute__((always_inline)) {
            if (G / NT >= STEPS) return;
            if (G % NT == 0) an = dv[G % RING]; else an += hv[G % RING] * dv[G % RING];
        };
        auto wfrag = [&](int grp) __attribute__((always_inline)) {         // A fragment of (step grp / TN, column group grp % TN)
            if (grp < STEPS * TN) wq[grp & 1] = Wl[2 * (grp / TN) * Cout + wrow[grp % TN]];
        };
        wfrag(0);
#pragma unroll
        for (int G = 0; G < RA; ++G) issue(G);
#pragma unroll
        for (int G = 0; G < NT; ++G) { issue(G + RA); consume(G); }         // step 0's depthwise: nothing to hide behind yet
        v4f a;
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fmaxf(an[e], dw_floor);
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const int grp = j * TN + sl / 4;
                if (sl % 4 == 0) wfrag(grp + 1);
                acc[sl / 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[grp & 1][sl % 4], a[sl % 4], acc[sl / 4], 0, 0, 0);
#pragma unroll
                for (int G = NT * (j + 1) + sl * NT / NS; G < NT * (j + 1) + (sl + 1) * NT / NS; ++G) { issue(G + RA); consume(G); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        DWPW_STAMP(4)
        // ---- epilogue: lane = pixel, accumulator quads = 4 consecutive channels.
        // (Stride-2 form, 320x320x16 -> 160x160x40: the phase stamps show stores 6.3 k + next prefetch's issue 4.1 k of 16.4 k cycles per
        // tile — queueing behind the memory pipeline.  Parking the wave's pixels in the halo rows it owns exclusively (4 wid + 1 .. + 3) and
        // writing them back as whole lines changed nothing (296 vs 301 us, stores still 5.9 k): it is the 4.6 TB/s of mixed read / write
        // traffic itself, not the 16-byte pieces, that the block waits for.)
        const int oy = ty0 + py, ox = tx0 + px;
        if (oy < p.Ho && ox < p.Wo) {
            float* __restrict__ orow = p.out1 + (((size_t)n * p.Ho + oy) * p.Wo + ox) * Cout;
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * jn + 8 * g + 4 * h;
                    if (co >= Cout) continue;
                    v4f v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[jn][4 * g + c], out_floor);
                    *reinterpret_cast<v4f*>(orow + co) = v;
                }
        }
        DWPW_STAMP(5)
    }
#ifdef FACEHIP_DWPW_PROF
    if (lane == 0 && p.slabs) {                                             // [workgroup][wave][8]: 6 phase sums, tile count
        long long* o = reinterpret_cast<long long*>(p.slabs) + ((size_t)blockIdx.x * 4 + wid) * 8;
        for (int i = 0; i < 6; ++i) o[i] = ph[i];
        o[6] = ntiles;
        o[7] = (__builtin_readcyclecounter() - clk0) * 1000 / ((long long)__builtin_amdgcn_s_memrealtime() - rt0 + 1);
    }
#endif
}

// ---- stride-2 block with more channels than one halo image can hold (SCRFD: 160x160x40 -> 80x80x72).  A 17 x 33 halo of all 40 channels is
// 99 KB — one workgroup per CU — so the K loop is split: channels [0, 4 CQA) and [4 CQA, 4 (CQA + CQB)) take turns in ONE halo buffer of
// 4 CQA channels per pixel (63 KB at CQA = 6: two workgroups per CU).  Per tile: part A's prefetched registers -> LDS, part A's steps, part B's
// registers -> LDS (same buffer), part B's steps, stores; the accumulators live through both.  Both parts' halos of the NEXT tile are in
// flight during this tile's work (A from the top of the tile, B from its middle).  Everything else is dwpw_reg_kernel: lane = (pixel, half),
// depthwise result -> MFMA B fragment, hand-scheduled slots, buffer loads with hardware zero fill, persistent workgroups, priority rotation.
template <int CQA, int CQB, int TN, int OCC>
__global__ __launch_bounds__(256, OCC) void dwpw_reg2_kernel(const ConvArgs p, const int tiles_x, const int tiles_y, const int tiles_total) {
    static_assert(CQA % 2 == 0 && CQB % 2 == 0 && CQB <= CQA, "whole 8-channel MFMA steps per part; part B fits part A's buffer");
    constexpr int DS = 2, CQ = CQA + CQB, C = CQ * 4, STEPS = C / 8, SA = CQA / 2;
    constexpr int PQ = CQA + 1;                                            // halo pixel pitch in float4 (odd)
    constexpr int HH = (DP_TH - 1) * DS + 3, HWD = (DP_TW - 1) * DS + 3, HALO = HH * HWD;   // 17 x 33 = 561
    constexpr int NPA = (HALO * CQA + 255) / 256, NPB = (HALO * CQB + 255) / 256;
    constexpr int HW2 = (HWD + 1) / 2;                                     // column planes of the stride-2 halo: see dwpw_reg_kernel
    auto slot = [](int hp) { const int hy = hp / HWD, hx = hp - hy * HWD; return hy * HWD + (hx & 1) * HW2 + (hx >> 1); };
    auto tapoff = [](int ky, int kx) { return ky * HWD + (kx & 1) * HW2 + (kx >> 1); };
    extern __shared__ v4f smem[];
    v4f* const halo = smem;                                                // [HALO][PQ]
    v4f* const dwl = halo + HALO * PQ;                                     // [10][CQ]: 9 taps + bias
    v4f* const Wl = dwl + 10 * CQ;                                         // [STEPS][2][Cout]
    float* const pwb = reinterpret_cast<float*>(Wl + STEPS * 2 * p.Cout);  // [32 * TN]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int Cout = p.Cout;
    const float dw_floor = p.dw_act == (int)Act::RELU ? 0.f : -INFINITY, out_floor = p.act == (int)Act::RELU ? 0.f : -INFINITY;

    for (int i = tid; i < 10 * CQ; i += 256) {
        const int k = i / CQ, q = i - k * CQ;
        dwl[i] = *reinterpret_cast<const v4f*>(k < 9 ? p.dw_w + (size_t)k * C + 4 * q : p.dw_b + 4 * q);
    }
    for (int i = tid; i < STEPS * 2 * Cout; i += 256) {
        const int n = i % Cout, jh = i / Cout;
        Wl[i] = *reinterpret_cast<const v4f*>(p.wt + (size_t)n * p.Kpad + 4 * jh);
    }
    for (int i = tid; i < 32 * TN; i += 256) pwb[i] = i < Cout ? p.bias[i] : 0.f;

    const int pix = wid * 32 + r, py = pix / DP_TW, px = (py & 1) ? (pix - py * DP_TW - 2) & (DP_TW - 1) : pix - py * DP_TW;   // (rotation: see dwpw_reg_kernel)
    const v4f* const hbase = halo + (py * DS * HWD + px) * PQ + h;         // + tapoff(ky, kx) * PQ + 2 (j - first step of the part)
    const v4f* const dbase = dwl + h;
    int wrow[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) wrow[jn] = h * Cout + min(32 * jn + r, Cout - 1);

    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, wgs = gridDim.x >> 3;
    const int q8 = tiles_total >> 3, r8 = tiles_total & 7;
    const int run0 = xcd * q8 + min(xcd, r8), run1 = run0 + q8 + (xcd < r8 ? 1 : 0);
    const int per_img = tiles_x * tiles_y;

    v4f pfa[NPA], pfb[NPB];
    int voa[NPA], vob[NPB];
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
        const int i = min(tid + 256 * k, HALO * CQA - 1);
        const int hp = i / CQA, q = i - hp * CQA;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        voa[k] = ((hy * p.W + hx) * C + 4 * q) * 4;
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
        const int i = min(tid + 256 * k, HALO * CQB - 1);
        const int hp = i / CQB, q = i - hp * CQB;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        vob[k] = ((hy * p.W + hx) * C + 4 * (CQA + q)) * 4;
    }
    const int img_bytes = p.H * p.W * C * 4;
    // (one part of tile t's halo: global -> registers, issued and not waited for; rows outside the image are out of the descriptor's range and
    //  read as zero, columns left / right of it are pushed out of range on the first / last tile column)
#define DWPW2_PREFETCH(PF, VO, NP, CQP, T)                                                                                                  \
    {                                                                                                                                       \
        const int n_ = (T) / per_img, rem_ = (T) - n_ * per_img;                                                                            \
        const int tyi_ = rem_ / tiles_x, txi_ = rem_ - tyi_ * tiles_x;                                                                      \
        const int y0_ = tyi_ * DP_TH * DS - 1, x0_ = txi_ * DP_TW * DS - 1;                                                                 \
        const __amdgpu_buffer_rsrc_t rsrc_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + (size_t)n_ * p.H * p.W * C), 0, img_bytes, 0x00020000); \
        const int tile_off_ = (y0_ * p.W + x0_) * C * 4;                                                                                    \
        int vo_[NP];                                                                                                                        \
        _Pragma("unroll") for (int k = 0; k < NP; ++k) vo_[k] = tile_off_ + VO[k];                                                          \
        if (x0_ < 0 || x0_ + HWD > p.W) {                                                                                                   \
            _Pragma("unroll") for (int k = 0; k < NP; ++k) {                                                                                \
                const int i = min(tid + 256 * k, HALO * CQP - 1);                                                                           \
                const int hx = (i / CQP) % HWD;                                                                                             \
                vo_[k] = (unsigned)(x0_ + hx) < (unsigned)p.W ? vo_[k] : (int)0x80000000;                                                   \
            }                                                                                                                               \
        }                                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < NP; ++k) PF[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc_, vo_[k], 0, 0)); \
    }
    int t = run0 + wg;
    if (t < run1) { DWPW2_PREFETCH(pfa, voa, NPA, CQA, t) DWPW2_PREFETCH(pfb, vob, NPB, CQB, t) }
    front_barrier();
    int prio_it = (int)(blockIdx.x >> 8);
    for (; t < run1; t += wgs) {
        if (!p.no_prio) rotate_wave_priority(prio_it++);
        const int n = t / per_img, rem = t - n * per_img;
        const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
        const int ty0 = tyi * DP_TH, tx0 = txi * DP_TW;
        v16f acc[TN];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f b = *reinterpret_cast<const v4f*>(pwb + 32 * jn + 8 * g + 4 * h);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[jn][4 * g + c] = b[c];
            }
        constexpr int NS = 4 * TN, NT = 10, RA = 3, RING = 4;              // (a ring of 5 spills 34 registers beside the two prefetch sets)
        // the K loop of one part: steps [J0, J1) of the block, whose channels are columns 2 (j - J0) + h of the halo now in LDS
        auto kpart = [&](auto j0c, auto j1c) __attribute__((always_inline)) {
            constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value, NJ = J1 - J0;
            v4f hv[RING], dv[RING], wq[2];
            auto issue = [&](int G) __attribute__((always_inline)) {       // LDS reads of the part's tap-slot G = NT * (step - J0) + t
                const int jl = G / NT, tt = G % NT;
                if (jl >= NJ) return;
                dv[G % RING] = dbase[(tt == 0 ? 9 : tt - 1) * CQ + 2 * (J0 + jl)];
                if (tt > 0) hv[G % RING] = hbase[tapoff((tt - 1) / 3, (tt - 1) % 3) * PQ + 2 * jl];
            };
            v4f an;
            auto consume = [&](int G) __attribute__((always_inline)) {
                if (G / NT >= NJ) return;
                if (G % NT == 0) an = dv[G % RING]; else an += hv[G % RING] * dv[G % RING];
            };
            auto wfrag = [&](int grp) __attribute__((always_inline)) {     // A fragment of (step J0 + grp / TN, column group grp % TN)
                if (grp < NJ * TN) wq[grp & 1] = Wl[2 * (J0 + grp / TN) * Cout + wrow[grp % TN]];
            };
            wfrag(0);
#pragma unroll
            for (int G = 0; G < RA; ++G) issue(G);
#pragma unroll
            for (int G = 0; G < NT; ++G) { issue(G + RA); consume(G); }
            v4f a;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = fmaxf(an[e], dw_floor);
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const int grp = j * TN + sl / 4;
                    if (sl % 4 == 0) wfrag(grp + 1);
                    acc[sl / 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[grp & 1][sl % 4], a[sl % 4], acc[sl / 4], 0, 0, 0);
#pragma unroll
                    for (int G = NT * (j + 1) + sl * NT / NS; G < NT * (j + 1) + (sl + 1) * NT / NS; ++G) { issue(G + RA); consume(G); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // ---- part A
        front_barrier();                                                    // every wave is done reading the previous tile's part-B halo
#pragma unroll
        for (int k = 0; k < NPA; ++k) {
            const int i = tid + 256 * k;
            if (i < HALO * CQA) halo[slot(i / CQA) * PQ + i % CQA] = pfa[k];
        }
        if (t + wgs < run1) DWPW2_PREFETCH(pfa, voa, NPA, CQA, t + wgs)
        front_barrier();
        kpart(std::integral_constant<int, 0>{}, std::integral_constant<int, SA>{});
        // ---- part B (same buffer)
        front_barrier();                                                    // every wave is done reading part A
#pragma unroll
        for (int k = 0; k < NPB; ++k) {
            const int i = tid + 256 * k;
            if (i < HALO * CQB) halo[slot(i / CQB) * PQ + i % CQB] = pfb[k];
        }
        if (t + wgs < run1) DWPW2_PREFETCH(pfb, vob, NPB, CQB, t + wgs)
        front_barrier();
        kpart(std::integral_constant<int, SA>{}, std::integral_constant<int, STEPS>{});
        // ---- epilogue: lane = pixel, accumulator quads = 4 consecutive channels
        const int oy = ty0 + py, ox = tx0 + px;
        if (oy < p.Ho && ox < p.Wo) {
            float* __restrict__ orow = p.out1 + (((size_t)n * p.Ho + oy) * p.Wo + ox) * Cout;
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * jn + 8 * g + 4 * h;
                    if (co >= Cout) continue;
                    v4f v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(acc[jn][4 * g + c], out_floor);
                    *reinterpret_cast<v4f*>(orow + co) = v;
                }
        }
    }
#undef DWPW2_PREFETCH
}

// Measured and not kept (round 3): the register-fed form WITHOUT a halo image for the stride-2 blocks (every lane gathers the nine taps of
// its own pixel straight from memory into the ring registers, no barrier in the tile loop at all): 496 / 369 us against 322 / 255 us
// of dwpw_kernel<.., 2, .., true> on 320x320x16 -> 160x160x40 / 160x160x40 -> 80x80x72.  With lane = pixel a wave-instruction touches 64
// different 128-byte lines (16 bytes each, stride 2 pixels): the loads are bound by tag look-ups, not by bytes.  The strip form of
// dwpw_kernel (lane = 16-byte channel column, 4 lanes per 64-byte pixel) keeps them coalesced and stays.

static size_t dwpw_reg_lds(int CQ, int Cout, int TN, int DS) {
    const int halo = ((DP_TH - 1) * DS + 3) * ((DP_TW - 1) * DS + 3);
    return ((size_t)halo * (CQ + 1) + 10 * CQ + (size_t)(CQ / 2) * 2 * Cout) * 16 + (size_t)32 * TN * 4;
}
static bool dwpw_reg_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_DWPW_REG"); v = e ? atoi(e) : 1; }
    return v != 0;
}
template <int CQ, int TN, int OCC, int DS = 1>
static void launch_dwpw_reg_cfg(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.Wo + DP_TW - 1) / DP_TW, tiles_y = (a.Ho + DP_TH - 1) / DP_TH;
    const int tiles_total = a.B * tiles_y * tiles_x;
    const int cus = a.cus > 0 ? a.cus : conv_num_cus();
    const size_t lds = dwpw_reg_lds(CQ, a.Cout, TN, DS);
    static bool attr_set = false;                                          // (per instantiation) dynamic LDS beyond the 64 KB default needs the opt-in
    if (!attr_set) {
        FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpw_reg_kernel<CQ, TN, OCC, DS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int grid = std::min((tiles_total + 7) / 8 * 8, cus * OCC);
    grid = std::max(8, grid / 8 * 8);
    ConvArgs ap = a;
    { static int pr = -1; if (pr < 0) { const char* e = getenv("FACEHIP_DWPW_PRIO"); pr = e ? atoi(e) : 1; } ap.no_prio = pr ? 0 : 1; }   // (0: no priority rotation — A / B timing)
    hipLaunchKernelGGL((dwpw_reg_kernel<CQ, TN, OCC, DS>), dim3((unsigned)grid), dim3(256), lds, s, ap, tiles_x, tiles_y, tiles_total);
}
template <int CQA, int CQB, int TN, int OCC>
static void launch_dwpw_reg2_cfg(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.Wo + DP_TW - 1) / DP_TW, tiles_y = (a.Ho + DP_TH - 1) / DP_TH;
    const int tiles_total = a.B * tiles_y * tiles_x;
    const int cus = a.cus > 0 ? a.cus : conv_num_cus();
    constexpr int CQ = CQA + CQB;
    const size_t lds = ((size_t)(7 * 2 + 3) * (15 * 2 + 3) * (CQA + 1) + 10 * CQ + (size_t)(CQ / 2) * 2 * a.Cout) * 16 + (size_t)32 * TN * 4;
    static bool attr_set = false;
    if (!attr_set) {
        FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpw_reg2_kernel<CQA, CQB, TN, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int grid = std::min((tiles_total + 7) / 8 * 8, cus * OCC);
    grid = std::max(8, grid / 8 * 8);
    ConvArgs ap = a;
    { static int pr = -1; if (pr < 0) { const char* e = getenv("FACEHIP_DWPW_PRIO"); pr = e ? atoi(e) : 1; } ap.no_prio = pr ? 0 : 1; }
    hipLaunchKernelGGL((dwpw_reg2_kernel<CQA, CQB, TN, OCC>), dim3((unsigned)grid), dim3(256), lds, s, ap, tiles_x, tiles_y, tiles_total);
}
// true = launched.  Instantiated for SCRFD-500M's stride-1 blocks (C = 16 / 40 / 64 / 72) with any Cout <= 96.
static bool launch_dwpw_reg(const ConvArgs& a, hipStream_t s) {
    if (!dwpw_reg_enabled() || a.u8_src || a.Cout % 4 || a.Cout > 96) return false;
    if (a.act != (int)Act::NONE && a.act != (int)Act::RELU) return false;
    if ((long)a.H * a.W * a.Cin * 4 >= (1L << 31)) return false;            // one image must fit a 32-bit buffer range
    const int tn = (a.Cout + 31) / 32;
    if (a.dw_stride == 2) {
        // the stride-2 block of 16 channels (SCRFD: 320x320x16 -> 160x160x40): a 17 x 33 halo of all channels is 45 KB, three workgroups per
        // CU; with 40 channels it would be 99 KB (one workgroup per CU) — those blocks stay with dwpw_kernel
        static int s2 = -1;
        if (s2 < 0) { const char* e = getenv("FACEHIP_DWPW_REG_S2"); s2 = e ? atoi(e) : 1; }
        if (!s2 || a.Ho != (a.H - 1) / 2 + 1 || a.Wo != (a.W - 1) / 2 + 1) return false;
        if (a.Cin == 40 && tn == 3) { launch_dwpw_reg2_cfg<6, 4, 3, 2>(a, s); return true; }      // (K split: 24 + 16 channels through one 63 KB halo buffer)
        if (a.Cin != 16 || tn > 2) return false;
        if (tn == 1) launch_dwpw_reg_cfg<4, 1, 3, 2>(a, s); else launch_dwpw_reg_cfg<4, 2, 3, 2>(a, s);
        return true;
    }
    if (a.dw_stride != 1 || a.H != a.Ho || a.W != a.Wo) return false;
    switch (a.Cin) {
        case 16:
            if (tn == 1) launch_dwpw_reg_cfg<4, 1, 4>(a, s); else if (tn == 2) launch_dwpw_reg_cfg<4, 2, 4>(a, s); else launch_dwpw_reg_cfg<4, 3, 3>(a, s);
            return true;
        case 40:
            if (tn == 1) launch_dwpw_reg_cfg<10, 1, 3>(a, s); else if (tn == 2) launch_dwpw_reg_cfg<10, 2, 3>(a, s); else launch_dwpw_reg_cfg<10, 3, 3>(a, s);
            return true;
        case 64:
            if (tn == 1) launch_dwpw_reg_cfg<16, 1, 2>(a, s); else if (tn == 2) launch_dwpw_reg_cfg<16, 2, 2>(a, s); else launch_dwpw_reg_cfg<16, 3, 2>(a, s);
            return true;
        case 72:
            if (tn == 1) launch_dwpw_reg_cfg<18, 1, 2>(a, s); else if (tn == 2) launch_dwpw_reg_cfg<18, 2, 2>(a, s); else launch_dwpw_reg_cfg<18, 3, 2>(a, s);
            return true;
        default: return false;
    }
}

bool front_fused_ok(int Cin, int Cout, int dw_stride) { return Cin == 16 && dw_stride == 1 && Cout <= 32 && Cout % 4 == 0; }

static void launch_front(const ConvArgs& a, hipStream_t s) {
    if (!front_fused_ok(a.Cin, a.Cout, a.dw_stride) || a.Kpad != 32 || (a.u8_stride != 1 && a.u8_stride != 2) || !a.stem_wfrag)
        throw std::runtime_error("dwpw: the fused stem needs 16 channels, stride 1 and Cout <= 32 (multiple of 4)");
    if ((long)a.u8_srcH * a.u8_step >= (1L << 31)) throw std::runtime_error("dwpw: frame too large for 32-bit byte offsets");
    const int tiles_x = (a.Wo + DP_TW - 1) / DP_TW, tiles_y = (a.Ho + DP_TH - 1) / DP_TH;
    const int tiles_total = a.B * tiles_y * tiles_x;
    const int cus = a.cus > 0 ? a.cus : conv_num_cus();
    int grid = std::min((tiles_total + 7) / 8 * 8, cus * 4);              // persistent: 4 workgroups per CU, a multiple of 8 (XCDs)
    grid = std::max(8, grid / 8 * 8);
    // A workgroup walks tiles t, t + grid / 8, ...: with a stride that shares a factor with the tile grid (128 and 20 columns: period 5)
    // some workgroups meet a border COLUMN in every fifth tile and others never — border tiles run the per-byte pass and cost ~2x, and the
    // phase stamps showed the slowest wave 33 % above the mean.  A stride coprime with tiles_x and tiles_y gives everyone the same mix.
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    int wgs = grid / 8;
    while (wgs > 1 && (gcd(wgs, tiles_x) != 1 || gcd(wgs, tiles_y) != 1)) --wgs;
    if (wgs * 8 * 10 >= grid * 9) grid = wgs * 8;                           // (only if it costs < 10 % of the workgroups)
    const dim3 g3((unsigned)grid);
    ConvArgs ap = a;
    { static int pr = -1; if (pr < 0) { const char* e = getenv("FACEHIP_FRONT_PRIO"); pr = e ? atoi(e) : 1; } ap.no_prio = pr ? 0 : 1; }   // (0: no priority rotation — A / B timing)
    if (a.u8_step % 4 == 0) hipLaunchKernelGGL(front_kernel<true>, g3, dim3(256), 0, s, ap, tiles_x, tiles_y, tiles_total);
    else hipLaunchKernelGGL(front_kernel<false>, g3, dim3(256), 0, s, ap, tiles_x, tiles_y, tiles_total);
}

void launch_dwpw(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.zeros = conv_zero_line();
    if ((long)a.B * a.Ho * a.Wo <= 0) return;
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    if (a.u8_src) launch_front(a, s);
    else if (launch_dwpw_reg(a, s)) {}
    else if (a.Cout <= 32) launch_dwpw_cfg<32, 4, 1>(a, s);
    else if (a.Cout <= 64) launch_dwpw_cfg<64, 2, 2>(a, s);
    else if (a.Cout <= 96) launch_dwpw_cfg<96, 4, 1>(a, s);
    else launch_dwpw_cfg<128, 2, 2>(a, s);
    timer.end(s, 4, a.t_flops, a.t_bytes);
}

}  // namespace fh
