// winograd.hip — Winograd F(4x4, 3x3) for the deep 3x3 stride-1 convolutions (Cin >= 128) of IResNet.
//
// What it replaces: the same Conv nodes conv_mfma.hip computes directly (reference: ORT inside session_->Run,
// src/face_recognizer.cpp:279-283).  Y = A^T [ (G g G^T) (.) (B^T d B) ] A on 4x4 output tiles: 36 multiplies per
// 16 outputs and input channel instead of 144 — the matrix-core work of these layers drops 4x, at the price of two
// bandwidth-bound transform passes and of fp32 rounding errors ~25x those of the direct form (measured in
// DESIGN.md; far inside the path's tolerance, but it is why the direct kernel stays the default for thin layers
// and why the switch fh_rec_set_winograd exists).
//
//   wino_input_kernel   d (6x6 input patch per tile, zero padded)  ->  V[f][tile][ci] = B^T d B, f = 6*i + j
//   wino_gemm_kernel    36 independent GEMMs in ONE launch:  M[f] = V[f] (tiles x Cin) * U[f] (Cin x Cout)
//                       (conv_igemm_kernel's grouped instantiation when Cout is off the 32 grid)
//   wino_output_kernel  Y = A^T M A, + bias -> PReLU / ReLU -> (+ residual) -> out, optional second output y*s2+t2
//
// U[f] = G g G^T is computed once at load time in fp64 (engine.cpp).  Interpolation points 0, +-1, +-2, inf
// (Lavin & Gray 2016).  Thread = one tile x 4 channels; every load / store is a 16-byte lane access and
// consecutive lanes walk the channels, so all transfers are whole 128-byte lines.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

#include "kernels.h"
#include "plan.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Split-bf16 operand format of the opt-in "bf16x2" mode (fh_rec_set_precision): a value x travels as ONE 32-bit word holding
// hi = bf16(x) in the low half and mid = bf16(x - hi) in the high half — 16 mantissa bits, same bytes as fp32, so the V / U buffers,
// their indexing and the GEMM's LDS-DMA loader do not change at all; only the matrix instruction does (three bf16 products
// hh + hm + mh with f32 accumulation instead of one f32 product).
__device__ __forceinline__ v4f wino_pack_bf16x2(const v4f x) {
    v4u out;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const v2f a = {x[2 * p], x[2 * p + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));        // v_cvt_pk_bf16_f32 (round to nearest even)
        const v2f r = {a[0] - __builtin_bit_cast(float, hb << 16), a[1] - __builtin_bit_cast(float, hb & 0xffff0000u)};   // exact
        const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
        out[2 * p] = (hb & 0xffffu) | (mb << 16);
        out[2 * p + 1] = (hb >> 16) | (mb & 0xffff0000u);
    }
    return __builtin_bit_cast(v4f, out);
}

__global__ __launch_bounds__(256) void pack_bf16x2_kernel(const float* __restrict__ in, float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        *reinterpret_cast<v4f*>(out + 4 * i) = wino_pack_bf16x2(*reinterpret_cast<const v4f*>(in + 4 * i));
}
void launch_pack_bf16x2(const float* in, float* out, long n, hipStream_t s) {       // n % 4 == 0
    const long n4 = n / 4;
    if (n4 > 0) hipLaunchKernelGGL(pack_bf16x2_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 256 * 64)), dim3(256), 0, s, in, out, n4);
}

// B^T (6x6) applied to a 6-vector
__device__ __forceinline__ void wino_bt(const v4f (&d)[6], v4f (&t)[6]) {
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
    t[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
    t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// A^T (4x6) applied to a 6-vector
__device__ __forceinline__ void wino_at(const v4f (&m)[6], v4f (&y)[4]) {
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
    y[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
    y[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}

// in [B,H,W,C] -> V [36][NT][C], NT = B*TY*TX tiles of 4x4 outputs (input patch rows 4ty-1 .. 4ty+4)
// aff_s / aff_t (optional): per-channel affine applied to the in-image pixels only — the block's pre-conv BatchNorm, so that the
// producer need not write a normalised copy of its output (zero padding stays exactly zero).
__global__ __launch_bounds__(256, 2) void wino_input_kernel(const float* __restrict__ in, float* __restrict__ V, int B, int H, int W,
                                                            int C, int TY, int TX, long NTp, const float* __restrict__ aff_s,
                                                            const float* __restrict__ aff_t, const int pack) {
    const int C4 = C >> 2;
    // (32-bit index arithmetic: tiles * C/4 < 2^31 is checked by the launcher; 64-bit div / mod per thread is not free)
    const int NT = B * TY * TX;
    const int total = NT * C4;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int c4 = idx % C4;
        const int tile = idx / C4;
        const int tx = tile % TX, ty = (tile / TX) % TY, b = tile / (TX * TY);
        const int iy0 = 4 * ty - 1, ix0 = 4 * tx - 1;
        const float* img = in + (size_t)b * H * W * C + c4 * 4;
        v4f as4 = {1.f, 1.f, 1.f, 1.f}, at4 = {0.f, 0.f, 0.f, 0.f};
        if (aff_s) { as4 = *reinterpret_cast<const v4f*>(aff_s + c4 * 4); at4 = *reinterpret_cast<const v4f*>(aff_t + c4 * 4); }
        v4f t[6][6];                                         // t[i][c] = (B^T d)[i][c]: columns of the patch first
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int ix = ix0 + c;
            v4f d[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int iy = iy0 + r;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                // UNCONDITIONAL load (out-of-image taps read pixel 0 and are zeroed afterwards): a load under a branch is waited for at
                // the branch's join — 36 serialised memory latencies per thread instead of 36 loads in flight
                const v4f raw = *reinterpret_cast<const v4f*>(img + (ok ? ((size_t)iy * W + ix) * C : (size_t)0));
                const float okf = ok ? 1.f : 0.f;                    // (a select on the RESULT would let the compiler sink the load back under a branch)
                d[r] = raw * (as4 * okf) + at4 * okf;
            }
            v4f tc[6];
            wino_bt(d, tc);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][c] = tc[i];
        }
        float* dst = V + (size_t)tile * C + c4 * 4;
        const size_t fs = (size_t)NTp * C;                   // stride between frequency planes (rows padded to whole GEMM tiles)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            v4f o[6];
            wino_bt(t[i], o);                                // (B^T d B)[i][j] = sum_c (B^T d)[i][c] * B^T[j][c]
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<v4f*>(dst + (size_t)(i * 6 + j) * fs) = pack ? wino_pack_bf16x2(o[j]) : o[j];
        }
    }
}

struct WinoOutArgs {
    const float* M; const float* bias; const float* slope; const float* res; float* out1; float* out2; const float* s2; const float* t2;
    int B, H, W, C, TY, TX, act;
    long NTp;
};

// M [36][NT][C] -> out [B,H,W,C] (H x W = output grid = input grid), epilogue as conv_mfma.hip's
__global__ __launch_bounds__(256, 2) void wino_output_kernel(const WinoOutArgs p) {
    const int C = p.C, C4 = C >> 2;
    const int NT = p.B * p.TY * p.TX;
    const int total = NT * C4;
    const size_t fs = (size_t)p.NTp * C;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int c4 = idx % C4;
        const int tile = idx / C4;
        const int tx = tile % p.TX, ty = (tile / p.TX) % p.TY, b = tile / (p.TX * p.TY);
        const float* src = p.M + (size_t)tile * C + c4 * 4;
        v4f t[4][6];                                         // t[y][j] = (A^T M)[y][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            v4f m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const v4f*>(src + (size_t)(i * 6 + j) * fs);
            v4f y[4];
            wino_at(m, y);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r][j] = y[r];
        }
        const v4f b4 = p.bias ? *reinterpret_cast<const v4f*>(p.bias + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
        v4f sl = {0.f, 0.f, 0.f, 0.f}, s2 = sl, t2 = sl;
        if (p.act == (int)Act::PRELU) sl = *reinterpret_cast<const v4f*>(p.slope + c4 * 4);
        if (p.out2) { s2 = *reinterpret_cast<const v4f*>(p.s2 + c4 * 4); t2 = *reinterpret_cast<const v4f*>(p.t2 + c4 * 4); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 4 * ty + r;
            if (oy >= p.H) continue;
            v4f y[4];
            wino_at(t[r], y);
            // the row's four residual reads BEFORE its stores: loads and stores share the in-order vmcnt, a residual read issued after a
            // store can only be waited for together with that store's acknowledgement (16 round trips per thread otherwise; all 16 reads
            // up front would not fit the register file here — the fused kernel does that)
            v4f rs[4];
            if (p.res) {
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    rs[x] = *reinterpret_cast<const v4f*>(p.res + (((size_t)b * p.H + oy) * p.W + min(4 * tx + x, p.W - 1)) * C + c4 * 4);   // (clamped: unconditional load)
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int ox = 4 * tx + x;
                if (ox >= p.W) continue;
                v4f v = y[x] + b4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u = v[e];
                    if (p.act == (int)Act::RELU) u = u > 0.f ? u : 0.f;
                    else if (p.act == (int)Act::PRELU) u = u >= 0.f ? u : u * sl[e];
                    else if (p.act == (int)Act::SIGMOID) u = 1.0f / (1.0f + expf(-u));
                    v[e] = u;
                }
                const size_t o = (((size_t)b * p.H + oy) * p.W + ox) * C + c4 * 4;
                if (p.res) v += rs[x];
                if (p.out1) *reinterpret_cast<v4f*>(p.out1 + o) = v;
                if (p.out2) *reinterpret_cast<v4f*>(p.out2 + o) = v * s2 + t2;
            }
        }
    }
}

// ---- output transform of conv L fused with the input transform of conv L+1 (same map, C = Cout(L) = Cin(L+1)) ---------------------
// Between two Winograd layers the activation only has to exist long enough to be re-tiled: 4x4 output tiles in, overlapping 6x6 input
// patches out.  One workgroup takes IMG whole images x 64 channels (14x14: one image, 7x7: four): phase 1 = wino_output_kernel's work
// (Y = A^T M A, bias -> PReLU / ReLU -> + residual; out1 / out2 written only if something else reads them), the values the next
// convolution sees — plain, or through its block's pre-conv BatchNorm (feed_aff) — go to an LDS image [pixel][64 channels]; phase 2 =
// wino_input_kernel's work from that image (zero padding = pixels outside the map).  The activation makes no round trip through memory
// and one launch disappears per layer (IResNet-50: 29 of the 38 Winograd layers feed another one on a <= 16x16 map).
struct WinoFuseArgs {
    WinoOutArgs o;
    float* V;               // [36][NTp][C] of the next convolution
    int feed_aff;           // 1: the next convolution reads y * s2 + t2 (its block's BatchNorm), 0: y itself
    // work split: a workgroup takes `img` whole images x (4 * csl4) channels; thread = (image, tile, float4 column).  Maps of <= 16 tiles
    // (14x14, 7x7): csl4 = 16 (64 channels, whole 256-byte pixel rows per 16 lanes); up to 64 tiles (28x28 = 49): csl4 = 4 (16 channels)
    // so that the LDS image of one picture still fits.  pitch = float4 slots per pixel in LDS (5 for csl4 = 4: bank spread).
    int csl4, pitch, img;
    int pack;               // the next convolution's GEMM takes split-bf16 operands
};

__global__ __launch_bounds__(256, 2) void wino_fused_kernel(const WinoFuseArgs a) {
    const WinoOutArgs& p = a.o;
    extern __shared__ v4f act[];                               // [img][H*W][pitch float4]
    const int C = p.C, TPI = p.TY * p.TX, IMG = a.img, HW = p.H * p.W, PITCH = a.pitch;
    const int tid = threadIdx.x, c4l = tid % a.csl4, slot = tid / a.csl4;
    const int cslices = C / (4 * a.csl4);
    const int cs = blockIdx.x % cslices, b0 = (blockIdx.x / cslices) * IMG;
    const int il = slot / TPI, tile = slot - il * TPI;
    const int ty = tile / p.TX, tx = tile - ty * p.TX;
    const int b = b0 + il;
    const bool live = il < IMG && b < p.B;
    const int c4 = cs * a.csl4 + c4l;                          // float4 index inside C
    const size_t fs = (size_t)p.NTp * C;
    const size_t tg = ((size_t)b * p.TY + ty) * p.TX + tx;    // tile row of the V / M planes
    if (live) {
        // ---- phase 1: Y = A^T M A, epilogue, LDS image
        const float* src = p.M + tg * C + c4 * 4;
        v4f t[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            v4f m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const v4f*>(src + (size_t)(i * 6 + j) * fs);
            v4f y[4];
            wino_at(m, y);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r][j] = y[r];
        }
        const v4f b4 = p.bias ? *reinterpret_cast<const v4f*>(p.bias + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
        v4f rs[4][4];                                        // residual reads before the first store (see wino_output_kernel)
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int oy = 4 * ty + r, ox = 4 * tx + x;
                    rs[r][x] = *reinterpret_cast<const v4f*>(p.res + (((size_t)b * p.H + min(oy, p.H - 1)) * p.W + min(ox, p.W - 1)) * C + c4 * 4);   // (clamped: unconditional)
                }
        }
        v4f sl = {0.f, 0.f, 0.f, 0.f}, s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        if (p.act == (int)Act::PRELU) sl = *reinterpret_cast<const v4f*>(p.slope + c4 * 4);
        if (p.s2) { s2 = *reinterpret_cast<const v4f*>(p.s2 + c4 * 4); t2 = *reinterpret_cast<const v4f*>(p.t2 + c4 * 4); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 4 * ty + r;
            if (oy >= p.H) continue;
            v4f y[4];
            wino_at(t[r], y);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int ox = 4 * tx + x;
                if (ox >= p.W) continue;
                v4f v = y[x] + b4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u = v[e];
                    if (p.act == (int)Act::RELU) u = u > 0.f ? u : 0.f;
                    else if (p.act == (int)Act::PRELU) u = u >= 0.f ? u : u * sl[e];
                    else if (p.act == (int)Act::SIGMOID) u = 1.0f / (1.0f + expf(-u));
                    v[e] = u;
                }
                const size_t o = (((size_t)b * p.H + oy) * p.W + ox) * C + c4 * 4;
                if (p.res) v += rs[r][x];
                if (p.out1) *reinterpret_cast<v4f*>(p.out1 + o) = v;
                const v4f vb = v * s2 + t2;
                if (p.out2) *reinterpret_cast<v4f*>(p.out2 + o) = vb;
                act[(il * HW + oy * p.W + ox) * PITCH + c4l] = a.feed_aff ? vb : v;
            }
        }
    }
    __syncthreads();
    if (!live) return;
    // ---- phase 2: V = B^T d B from the LDS image (6x6 patch, rows / columns 4t-1 .. 4t+4, zeros outside the map)
    const v4f* img = act + (size_t)il * HW * PITCH + c4l;
    const int iy0 = 4 * ty - 1, ix0 = 4 * tx - 1;
    v4f tt[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int ix = ix0 + c;
        v4f d[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = iy0 + r;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            d[r] = ok ? img[(iy * p.W + ix) * PITCH] : v4f{0.f, 0.f, 0.f, 0.f};
        }
        v4f tc[6];
        wino_bt(d, tc);
#pragma unroll
        for (int i = 0; i < 6; ++i) tt[i][c] = tc[i];
    }
    float* dst = a.V + tg * C + c4 * 4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v4f o6[6];
        wino_bt(tt[i], o6);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<v4f*>(dst + (size_t)(i * 6 + j) * fs) = a.pack ? wino_pack_bf16x2(o6[j]) : o6[j];
    }
}

// ---- the same fusion for maps of up to 56 tiles (28x28 = 49): one workgroup = ONE image x a 32-channel slice -------------------------
// wino_fused_kernel's 16-channel slices (csl4 = 4) turn every activation access — residual, out1 / out2 — into 64-byte pieces, which is
// why junctions that touch memory kept two separate kernels on these maps (116 us fused against 60 + 46 separate, 28x28x128 at B = 128).
// With 32 channels a slice of a pixel is one whole 128-byte line, and the whole 28 x 28 slice (100 KB) still fits the CU's 160 KB of LDS:
// thread = (tile, float4 column), 8 lanes per line, 7 waves per workgroup, one workgroup per CU.  LDS image [row][32 pixel slots][8 float4]:
// the slot of column x has bits 0 and 2 of x exchanged, so that the two tiles of a ds_read_b128 lane group (x and x + 4) sit in opposite
// halves of the 64 banks (a pixel's 128 bytes cover half of them) — without it every phase-2 read is a 2-way conflict.
constexpr int kSliceThreads = 448, kSliceTiles = kSliceThreads / 8;
__device__ __forceinline__ int wino_slice_slot(int x) { return (x & ~5) | ((x >> 2) & 1) | ((x & 1) << 2); }

__global__ __launch_bounds__(kSliceThreads) void wino_slice_kernel(const WinoFuseArgs a) {
    const WinoOutArgs& p = a.o;
    extern __shared__ v4f act[];                               // [H][32 slots][8 float4]
    const int C = p.C, TPI = p.TY * p.TX;
    const int tid = threadIdx.x, c4l = tid & 7, tile = tid >> 3;
    const int cslices = C >> 5;
    // workgroups b, b + 8, b + 16 ... share an XCD (round-robin dispatch): the slices of one image go to ONE XCD, so that the 128-byte
    // quarters of a 512-byte plane row are fetched by one L2 close together in time (speed only; any mapping is correct)
    int blk = blockIdx.x;
    if ((p.B & 7) == 0) { const int x = blk & 7, j = blk >> 3; blk = ((j / cslices) * 8 + x) * cslices + j % cslices; }
    const int cs = blk % cslices, b = blk / cslices;
    const int ty = tile / p.TX, tx = tile - ty * p.TX;
    const bool live = tile < TPI;
    const int c4 = cs * 8 + c4l;
    const size_t fs = (size_t)p.NTp * C;
    const bool slice_nt = a.img == 2;                          // (experiment switch FACEHIP_WINO_SLICE=2: M is read once and dead afterwards)
    const size_t tg = ((size_t)b * p.TY + ty) * p.TX + tx;
    if (live) {
        // ---- phase 1: Y = A^T M A, epilogue, LDS image (every global access of the 8 lanes of a tile is one 128-byte line)
        const float* src = p.M + tg * C + c4 * 4;
        v4f t[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            v4f m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = slice_nt ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + (size_t)(i * 6 + j) * fs)) : *reinterpret_cast<const v4f*>(src + (size_t)(i * 6 + j) * fs);
            v4f y[4];
            wino_at(m, y);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r][j] = y[r];
        }
        const v4f b4 = p.bias ? *reinterpret_cast<const v4f*>(p.bias + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
        v4f rs[4][4];                                        // residual reads before the first store (see wino_output_kernel)
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int oy = 4 * ty + r, ox = 4 * tx + x;
                    rs[r][x] = *reinterpret_cast<const v4f*>(p.res + (((size_t)b * p.H + min(oy, p.H - 1)) * p.W + min(ox, p.W - 1)) * C + c4 * 4);   // (clamped: unconditional)
                }
        }
        v4f sl = {0.f, 0.f, 0.f, 0.f}, s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        if (p.act == (int)Act::PRELU) sl = *reinterpret_cast<const v4f*>(p.slope + c4 * 4);
        if (p.s2) { s2 = *reinterpret_cast<const v4f*>(p.s2 + c4 * 4); t2 = *reinterpret_cast<const v4f*>(p.t2 + c4 * 4); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 4 * ty + r;
            if (oy >= p.H) continue;
            v4f y[4];
            wino_at(t[r], y);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int ox = 4 * tx + x;
                if (ox >= p.W) continue;
                v4f v = y[x] + b4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u = v[e];
                    if (p.act == (int)Act::RELU) u = u > 0.f ? u : 0.f;
                    else if (p.act == (int)Act::PRELU) u = u >= 0.f ? u : u * sl[e];
                    else if (p.act == (int)Act::SIGMOID) u = 1.0f / (1.0f + expf(-u));
                    v[e] = u;
                }
                const size_t o = (((size_t)b * p.H + oy) * p.W + ox) * C + c4 * 4;
                if (p.res) v += rs[r][x];
                if (p.out1) *reinterpret_cast<v4f*>(p.out1 + o) = v;
                const v4f vb = v * s2 + t2;
                if (p.out2) *reinterpret_cast<v4f*>(p.out2 + o) = vb;
                act[(oy * 32 + wino_slice_slot(ox)) * 8 + c4l] = a.feed_aff ? vb : v;
            }
        }
    }
    __syncthreads();
    if (!live) return;
    // ---- phase 2: V = B^T d B from the LDS image (6x6 patch, rows / columns 4t-1 .. 4t+4, zeros outside the map)
    const v4f* img = act + c4l;
    const int iy0 = 4 * ty - 1, ix0 = 4 * tx - 1;
    v4f tt[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int ix = ix0 + c;
        const int sx = wino_slice_slot(ix & 31);
        v4f d[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = iy0 + r;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            d[r] = ok ? img[(iy * 32 + sx) * 8] : v4f{0.f, 0.f, 0.f, 0.f};
        }
        v4f tc[6];
        wino_bt(d, tc);
#pragma unroll
        for (int i = 0; i < 6; ++i) tt[i][c] = tc[i];
    }
    float* dst = a.V + tg * C + c4 * 4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v4f o6[6];
        wino_bt(tt[i], o6);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<v4f*>(dst + (size_t)(i * 6 + j) * fs) = a.pack ? wino_pack_bf16x2(o6[j]) : o6[j];
    }
}

static int wino_slice_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_WINO_SLICE"); v = e ? atoi(e) : 1; }          // (0: the round-4 behaviour, for A / B timing)
    return v;
}
// maps of 17 .. 56 tiles whose 32-channel slice fits the LDS: W <= 32 (slot arithmetic), whole 32-channel slices
static bool wino_slice_shape(int H, int W, int C) {
    const int tpi = ((H + 3) / 4) * ((W + 3) / 4);
    return wino_slice_enabled() && C % 32 == 0 && tpi > 16 && tpi <= kSliceTiles && W <= 32 && (size_t)H * 32 * 8 * sizeof(v4f) <= 160 * 1024;
}

// ---- the 36 GEMMs: M[f] = V[f] (rows x K) * U[f] (K x N), frequency planes stacked along the rows --------------------------
// Same tile anatomy as conv_igemm_kernel (LDS-DMA with source-side swizzle, [row][32 k] LDS images, one ds_read_b128 per 4 MFMAs,
// weights as the MFMA A operand) but nothing else: rows are contiguous, K and N are multiples of 32, the row count a multiple of
// 256, the epilogue a plain store.  The generic kernel spends ~650 VALU + ~490 SALU instructions per wave on tap / padding / index
// bookkeeping around the 128 MFMAs of such a short-K tile (rocprofv3: matrix pipe busy 58 %); this one does not.
__device__ __forceinline__ void wino_dma16(const float* src, v4f* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

// weight matrix (frequency 6 i + j of the F(4x4) set) of the plane GEMM tile `tile_m` belongs to — see WinoPlanes (kernels.h)
__device__ __forceinline__ int wino_plane_freq(const WinoPlanes& pl, const int tile_m) {
    const int c = tile_m >= pl.e[3] ? 3 : tile_m >= pl.e[2] ? 2 : tile_m >= pl.e[1] ? 1 : 0;
    const int pln = (tile_m - pl.e[c]) / pl.t[c];
    const int nfc = pl.nfc[c], f2 = pl.f2[c];
    int i = pln / nfc, j = pln - i * nfc;
    if ((f2 & 2) && i == 3) i = 5;                            // F(2) points {0, 1, -1, inf} = F(4) frequencies {0, 1, 2, 5}
    if ((f2 & 1) && j == 3) j = 5;
    return 6 * i + j;
}

// A wave's 32 x (32 TN) accumulator block -> M as whole rows (see the comment at wino_gemm_kernel's stores): turned around in the wave's
// own region of `scratch` (the tile buffers, free after the K loop's last barrier), two 32-column blocks at a time.
template <int TN>
__device__ __forceinline__ void wino_store_lines(const v16f (&acc)[TN], float* scratch, size_t scratch_bytes, float* __restrict__ M, int N,
                                                 int m0, int n0, int wid, int lane) {
    constexpr int JB = TN >= 2 ? 2 : 1, W = JB * 32;                        // column blocks turned around at a time (BN = 128: two halves)
    constexpr int PITCH = W + 4, RPI = 64 / (W / 4);                        // floats per LDS row; rows per store instruction
    static_assert(TN % JB == 0, "store scratch");
    (void)scratch_bytes;
    const int fr = lane & 31, fh2 = lane >> 5;
    float* const blk = scratch + wid * 32 * PITCH;
    const int rr = lane / (W / 4), cq = lane % (W / 4);
#pragma unroll
    for (int h = 0; h < TN / JB; ++h) {
        wave_lds_order();                                                   // (the previous half's scratch reads lie above these writes)
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j = h * JB + jj;
                *reinterpret_cast<v4f*>(blk + fr * PITCH + jj * 32 + 8 * g + 4 * fh2) = v4f{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            }
        wave_lds_order();                                                   // (lane (rr, cq) reads rows other lanes wrote)
        float* const obase = M + (size_t)(m0 + wid * 32 + rr) * N + n0 + h * W + 4 * cq;
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i)
            *reinterpret_cast<v4f*>(obase + (size_t)i * RPI * N) = *reinterpret_cast<const v4f*>(blk + (rr + i * RPI) * PITCH + 4 * cq);
    }
}

// Round 5 measured two more arrangements of this loop, both bit-identical and both SLOWER, neither kept in the tree: the next chunk's 8 (6)
// LDS-DMA requests spread over the four k-steps between the MFMAs with the next step's fragments read a half-step ahead (GEMM sum of
// IResNet-50 at B = 128: 3 189-3 227 us against 3 127-3 160 in the same runs), and several tiles per workgroup (wino_gemm_pers_kernel below).
#ifdef FACEHIP_WINO_STAMP
// Diagnostic build (scripts/wino_gemm_stamps.sh): thread 0 of every workgroup of wino_gemm_kernel records the 100 MHz wall clock at entry,
// when the first chunk has landed, after the K loop and after its last store instruction, into a device buffer the script reads back.
__device__ unsigned long long g_wino_stamp[8 * 4096];
extern "C" __attribute__((visibility("default"))) int fh_debug_wino_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wino_stamp), sizeof(unsigned long long) * (size_t)(n < 8 * 4096 ? n : 8 * 4096)) == hipSuccess ? 0 : -1;
}
#define WINO_STAMP(i) if (tid == 0 && blockIdx.x < 4096) g_wino_stamp[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define WINO_STAMP(i)
#endif
// Phase stamps (scripts/wino_gemm_stamps.sh, profiles/r05_wino_gemm_stamps.txt; 14x14x256 at B = 128, 968 workgroups of 128x128, two per CU):
// first chunk 2.2 us, K loop 29.8 us for the two co-resident workgroups of a CU (27.3 at the MFMA peak: the loop itself runs at 0.92),
// turn-around + stores 2.6 us; the second round starts at 35-44 us with a 4.7 us prologue and a 25.6 us K loop — a workgroup ALONE on its CU
// does not run twice as fast (one wave per SIMD cannot cover its own chunk latency), which is why a half-tile start stagger of every CU's
// second workgroup measured +-0 (3 125-3 149 us GEMM sum at 3-9 us of stagger against 3 149-3 158 without).
// ABL (diagnostic instantiations only, results are garbage): 1 = no loads in the K loop (the first chunk is computed over and over),
// 2 = no LDS reads (operands stay in registers), 4 = no barriers in the K loop, 8 = no stores, 16 = stores straight from registers
template <int BN, int OCC, int ABL = 0>
__global__ __launch_bounds__(256, OCC) void wino_gemm_kernel(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M,
                                                            const int K, const int N, const WinoPlanes pl, const long wt_gs,
                                                            const int tiles_n, const int chunks) {
    constexpr int BM = 128, TN = BN / 32, AL = BM / 32, BL = BN / 32;
    __shared__ v4f lds[2][(BM + BN) * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    WINO_STAMP(0);
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;
    // XCD-contiguous tile order (blockIdx % 8 = XCD): the tiles_n workgroups that read the same rows share an L2
    const int nb = gridDim.x, q = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
    const int tile = x * q + min(x, r8) + (blockIdx.x >> 3);
    const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float* Ug = U + (size_t)wino_plane_freq(pl, tile_m) * wt_gs;

    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);              // source k-column of this lane (swizzle on the source side)
    const float* a_src = V + (size_t)(m0 + lrow) * K + lqs * 4;
    const float* b_src = Ug + (size_t)(n0 + lrow) * K + lqs * 4;
    const size_t row32 = (size_t)32 * K;
    v4f* const dstA = &lds[0][wid * 64];
    v4f* const dstB = &lds[0][BM * 8 + wid * 64];
    auto load_chunk = [&](int buf) {
        v4f* const dA = dstA + buf * ((BM + BN) * 8);
        v4f* const dB = dstB + buf * ((BM + BN) * 8);
#pragma unroll
        for (int i = 0; i < AL; ++i) wino_dma16(a_src + i * row32, dA + i * 32 * 8);
#pragma unroll
        for (int i = 0; i < BL; ++i) wino_dma16(b_src + i * row32, dB + i * 32 * 8);
        a_src += 32; b_src += 32;
    };
    v16f acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    load_chunk(0);
    __syncthreads();
    WINO_STAMP(1);
    v4f xk = v4f{0.f, 0.f, 0.f, 0.f}, wk[TN];
    if constexpr ((ABL & 2) != 0) {
        xk = lds[0][(wid * 32 + fr) * 8 + (fh2 ^ fsw)];
#pragma unroll
        for (int j = 0; j < TN; ++j) wk[j] = lds[0][BM * 8 + fr * 8 + j * 32 * 8 + (fh2 ^ fsw)];
    }
    for (int kc = 0; kc < chunks; ++kc) {
        const int buf = (ABL & 1) ? 0 : (kc & 1);
        if ((ABL & 1) == 0 && kc + 1 < chunks) load_chunk(buf ^ 1);
        const v4f* X = lds[buf] + (wid * 32 + fr) * 8;
        const v4f* Wt = lds[buf] + BM * 8 + fr * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int col = (2 * s + fh2) ^ fsw;
            v4f xv;
            v4f w[TN];
            if constexpr ((ABL & 2) != 0) {
                xv = xk;
#pragma unroll
                for (int j = 0; j < TN; ++j) w[j] = wk[j];
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(xv));                  // (opaque: the compiler must not fold the steps together)
#endif
            } else {
                xv = X[col];
#pragma unroll
                for (int j = 0; j < TN; ++j) w[j] = Wt[j * 32 * 8 + col];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], xv[e], acc[j], 0, 0, 0);
        }
        if constexpr ((ABL & 4) == 0) __syncthreads();
    }
    WINO_STAMP(2);
    if constexpr ((ABL & 8) != 0) { if (acc[0][0] != 123.456f) return; }
    // Stores.  A lane's accumulators are 4-column pieces of ITS row: stored straight from registers, one instruction writes 32 bytes to
    // each of 32 rows (a quarter of a 128-byte line each, 8 instructions per row).  Instead the wave turns its 32 x BN block around in
    // LDS (its own region of the tile buffers, which every wave has finished reading at the loop's last barrier; row pitch BN + 4 floats:
    // the 16 lanes of a b128 phase hit 64 different banks) and stores whole rows: 64 / (BN / 4) rows x BN * 4 contiguous bytes per instruction.
    if constexpr ((ABL & 16) == 0) {
        static_assert(4 * 32 * ((TN >= 2 ? 64 : 32) + 4) * sizeof(float) <= sizeof(lds), "store scratch");
        wino_store_lines<TN>(acc, reinterpret_cast<float*>(&lds[0][0]), sizeof(lds), M, N, m0, n0, wid, lane);
        WINO_STAMP(3);
#ifdef FACEHIP_WINO_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (diagnostic: when the stores are acknowledged)
        WINO_STAMP(4);
        if (tid == 0 && blockIdx.x < 4096) { g_wino_stamp[blockIdx.x * 8 + 5] = (unsigned long long)tile; g_wino_stamp[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((3 << 11) | 20); }
#endif
        return;
    }
    float* orow = M + (size_t)(m0 + wid * 32 + fr) * N + n0 + 4 * fh2;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<v4f*>(orow + j * 32 + 8 * g) = v4f{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
}

// ---- the same GEMM with several tiles per workgroup (round 5) ------------------------------------------------------------------------
// A launch of more tiles than resident workgroups runs in rounds, and in every round each workgroup pays the same two serial pieces around
// its K loop: the first chunk's trip from L2 / the Infinity Cache (nothing to compute meanwhile) and the store tail (the matrix pipe idles
// until the workgroup retires).  Here a workgroup walks tiles vb = blockIdx.x, + gridDim.x, ... and, once the K loop of a tile has passed its
// last barrier, requests the NEXT tile's first chunk into the tile buffer the turn-around scratch does not use, then stores: the DMA flies
// under the epilogue and the store acknowledgements under the next tile's first chunk.  The wait in front of the next K loop is COUNTED
// (`s_waitcnt vmcnt(stores issued since the DMA)`: vector-memory operations retire in issue order, so that covers the DMA and leaves the
// stores in flight) with a raw s_barrier — `__syncthreads()` would drain the stores too.  Results are bit-identical to wino_gemm_kernel
// (same tile arithmetic in the same order).  MEASURED AND NOT THE DEFAULT (see wino_pers_enabled): kept behind FACEHIP_WINO_PERS=1 as the
// record of the experiment the round-4 review asked for, with its own oracle test.
template <int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void wino_gemm_pers_kernel(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M,
                                                                 const int K, const int N, const WinoPlanes pl, const long wt_gs,
                                                                 const int tiles_n, const int chunks, const int ntiles) {
    constexpr int BM = 128, TN = BN / 32, AL = BM / 32, BL = BN / 32;
    constexpr int BUF = (BM + BN) * 8;                                      // float4 slots per tile buffer
    __shared__ v4f lds[2][BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;
    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);              // source k-column of this lane (swizzle on the source side)
    const size_t row32 = (size_t)32 * K;
    // XCD-contiguous tile order over the VIRTUAL block index (gridDim.x is a multiple of 8: vb % 8 = this workgroup's XCD for every vb)
    const int q = ntiles >> 3, r8 = ntiles & 7;
    auto tile_of = [&](int vb) { const int x = vb & 7; return x * q + min(x, r8) + (vb >> 3); };
    const float* a_src; const float* b_src;
    auto point_at = [&](int tile) {
        const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
        const float* Ug = U + (size_t)wino_plane_freq(pl, tile_m) * wt_gs;
        a_src = V + (size_t)(tile_m * BM + lrow) * K + lqs * 4;
        b_src = Ug + (size_t)(tile_n * BN + lrow) * K + lqs * 4;
    };
    auto load_chunk = [&](int buf) {
        v4f* const dA = &lds[0][wid * 64] + buf * BUF;
        v4f* const dB = &lds[0][BM * 8 + wid * 64] + buf * BUF;
#pragma unroll
        for (int i = 0; i < AL; ++i) wino_dma16(a_src + i * row32, dA + i * 32 * 8);
#pragma unroll
        for (int i = 0; i < BL; ++i) wino_dma16(b_src + i * row32, dB + i * 32 * 8);
        a_src += 32; b_src += 32;
    };
    constexpr int JB = 1, W = JB * 32, PITCH = W + 4, RPI = 64 / (W / 4);    // turn-around in 32-column blocks: 4 waves x 32 x 36 floats = 18 KB <= one buffer
    static_assert(4 * 32 * PITCH * sizeof(float) <= BUF * sizeof(v4f), "store scratch must fit ONE tile buffer");
    constexpr int NSTORE = TN * (32 / RPI);                                 // global store instructions per wave and tile
    int base = 0;                                                           // buffer that holds the current tile's chunk 0
    int vb = blockIdx.x;
    point_at(tile_of(vb));
    load_chunk(0);
    __syncthreads();
    for (;;) {
        const int tile = tile_of(vb);
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        v16f acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int kc = 0; kc < chunks; ++kc) {
            const int buf = (base + kc) & 1;
            if (kc + 1 < chunks) load_chunk(buf ^ 1);
            const v4f* X = lds[buf] + (wid * 32 + fr) * 8;
            const v4f* Wt = lds[buf] + BM * 8 + fr * 8;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int col = (2 * s4 + fh2) ^ fsw;
                const v4f xv = X[col];
                v4f w[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) w[j] = Wt[j * 32 * 8 + col];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], xv[e], acc[j], 0, 0, 0);
            }
            __syncthreads();
        }
        // every wave is past the last chunk: both buffers are free.  last = the buffer of the final chunk -> scratch; the other -> next tile
        const int last = (base + chunks - 1) & 1;
        const int nvb = vb + gridDim.x;
        const bool more = nvb < ntiles;
        if (more) {
            point_at(tile_of(nvb));
            load_chunk(last ^ 1);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);                              // (the DMA is issued in front of the stores it is counted against)
#endif
        }
        {
            float* const blk = reinterpret_cast<float*>(&lds[last][0]) + wid * 32 * PITCH;
            const int rr = lane / (W / 4), cq = lane % (W / 4);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                wave_lds_order();
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<v4f*>(blk + fr * PITCH + 8 * g + 4 * fh2) = v4f{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                wave_lds_order();
                float* const obase = M + (size_t)(m0 + wid * 32 + rr) * N + n0 + j * W + 4 * cq;
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i)
                    *reinterpret_cast<v4f*>(obase + (size_t)i * RPI * N) = *reinterpret_cast<const v4f*>(blk + (rr + i * RPI) * PITCH + 4 * cq);
            }
        }
        if (!more) break;
        vb = nvb; base = last ^ 1;
#if defined(__HIP_DEVICE_COMPILE__)
        // the next tile's chunk 0 has landed once all but the NSTORE youngest vector-memory operations (this tile's stores) are done
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTORE) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

// Split-bf16 form of the same GEMM (opt-in precision mode): V and U hold (hi, mid) bf16 pairs in 32-bit words, everything up to the
// fragment reads is byte-for-byte the f32 kernel.  A lane takes 8 consecutive k of its row per 16-deep MFMA step (two ds_read_b128),
// separates the hi and the mid halves with v_perm_b32 and issues v_mfma_f32_32x32x16_bf16 three times: mid*hi + hi*mid + hi*hi
// (the dropped mid*mid term is 2^-32 relative).  5x less matrix-core time per FLOP than v_mfma_f32_32x32x2_f32; the kernel is then bound
// by the L2 -> LDS stream it shares with the f32 form.
template <int BN, int OCC>
__global__ __launch_bounds__(256, OCC) void wino_gemm_bf16x2_kernel(const float* __restrict__ V, const float* __restrict__ U, float* __restrict__ M,
                                                                   const int K, const int N, const WinoPlanes pl, const long wt_gs,
                                                                   const int tiles_n, const int chunks) {
    constexpr int BM = 128, TN = BN / 32, AL = BM / 32, BL = BN / 32;
    __shared__ v4f lds[2][(BM + BN) * 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;
    const int nb = gridDim.x, q = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
    const int tile = x * q + min(x, r8) + (blockIdx.x >> 3);
    const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float* Ug = U + (size_t)wino_plane_freq(pl, tile_m) * wt_gs;
    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);
    const float* a_src = V + (size_t)(m0 + lrow) * K + lqs * 4;
    const float* b_src = Ug + (size_t)(n0 + lrow) * K + lqs * 4;
    const size_t row32 = (size_t)32 * K;
    v4f* const dstA = &lds[0][wid * 64];
    v4f* const dstB = &lds[0][BM * 8 + wid * 64];
    auto load_chunk = [&](int buf) {
        v4f* const dA = dstA + buf * ((BM + BN) * 8);
        v4f* const dB = dstB + buf * ((BM + BN) * 8);
#pragma unroll
        for (int i = 0; i < AL; ++i) wino_dma16(a_src + i * row32, dA + i * 32 * 8);
#pragma unroll
        for (int i = 0; i < BL; ++i) wino_dma16(b_src + i * row32, dB + i * 32 * 8);
        a_src += 32; b_src += 32;
    };
    auto split = [](const v4u lo4, const v4u hi4, bf16x8& h, bf16x8& m) {       // 8 packed words -> 8 hi halves, 8 mid halves
        const v4u hh = {__builtin_amdgcn_perm(lo4[1], lo4[0], 0x05040100u), __builtin_amdgcn_perm(lo4[3], lo4[2], 0x05040100u),
                        __builtin_amdgcn_perm(hi4[1], hi4[0], 0x05040100u), __builtin_amdgcn_perm(hi4[3], hi4[2], 0x05040100u)};
        const v4u mm = {__builtin_amdgcn_perm(lo4[1], lo4[0], 0x07060302u), __builtin_amdgcn_perm(lo4[3], lo4[2], 0x07060302u),
                        __builtin_amdgcn_perm(hi4[1], hi4[0], 0x07060302u), __builtin_amdgcn_perm(hi4[3], hi4[2], 0x07060302u)};
        h = __builtin_bit_cast(bf16x8, hh); m = __builtin_bit_cast(bf16x8, mm);
    };
    v16f acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    load_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < chunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < chunks) load_chunk(buf ^ 1);
        const v4u* X = reinterpret_cast<const v4u*>(lds[buf]) + (wid * 32 + fr) * 8;
        const v4u* Wt = reinterpret_cast<const v4u*>(lds[buf]) + BM * 8 + fr * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                             // two 16-deep steps per 32-deep chunk; lane half fh2 owns k = 8*fh2 .. 8*fh2+7
            const int c0 = (4 * ks + 2 * fh2) ^ fsw, c1 = (4 * ks + 2 * fh2 + 1) ^ fsw;
            bf16x8 xh, xm;
            split(X[c0], X[c1], xh, xm);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8 wh, wm;
                split(Wt[j * 32 * 8 + c0], Wt[j * 32 * 8 + c1], wh, wm);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    static_assert(4 * 32 * ((TN >= 2 ? 64 : 32) + 4) * sizeof(float) <= sizeof(lds), "store scratch");
    wino_store_lines<TN>(acc, reinterpret_cast<float*>(&lds[0][0]), sizeof(lds), M, N, m0, n0, wid, lane);
}

long wino_rows(long tiles) { return (tiles + 255) / 256 * 256; }   // rows of one frequency plane: whole tiles of every GEMM configuration

static inline int wino_grid(long n) {
    const long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 256 * 16 ? 256 * 16 : b);
}

// G g G^T of one 3x3 filter in fp64: g[ky][kx] -> u[36] (f = 6*i + j)
void wino_filter_transform(const double g[9], double u[36]) {
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    double t[6][3];
    for (int i = 0; i < 6; ++i)
        for (int kx = 0; kx < 3; ++kx) t[i][kx] = G[i][0] * g[0 * 3 + kx] + G[i][1] * g[1 * 3 + kx] + G[i][2] * g[2 * 3 + kx];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) u[i * 6 + j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
}

static void wino_check(long NT, int cmax) {
    if (NT * (cmax / 4) >= (1L << 31) || 36 * (NT + 256) >= (1L << 31))
        throw std::runtime_error("winograd: batch too large for the 32-bit tile index (split the batch)");
}

// stage 1: in [B,H,W,Cin] -> V (optional per-channel affine on in-image pixels)
void launch_wino_input(const ConvArgs& a, float* V, const float* in_scale, const float* in_shift, bool pack, hipStream_t s) {
    const int TY = (a.H + 3) / 4, TX = (a.W + 3) / 4;
    const long NT = (long)a.B * TY * TX;
    if (NT <= 0) return;
    wino_check(NT, std::max(a.Cin, a.Cout));
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL(wino_input_kernel, dim3(wino_grid(NT * (a.Cin >> 2))), dim3(256), 0, s, a.in, V, a.B, a.H, a.W, a.Cin, TY, TX, wino_rows(NT), in_scale,
                       in_shift, pack ? 1 : 0);
    timer.end(s, 8, 0.0, 0.0);
}

// stage 2: the 36 GEMMs  M[f] = V[f] * U[f]   (wt36 = 36 packed weight images [conv_wt_rows(Cout)][Cin])
// bf16x2: V and wt36 hold split-bf16 words (wino_gemm_ok_bf16x2 says whether this layer's GEMM has that form)
bool wino_gemm_ok_bf16x2(int Cin, int Cout) { return Cout % 64 == 0 && Cin % 32 == 0; }

static int g_wino_slots_override = 0;                           // test hook (fh_debug_wino_slots): pretend the chip has this many workgroup slots per launch
void wino_debug_slots(int slots) { g_wino_slots_override = slots > 0 ? (slots + 7) / 8 * 8 : 0; }
static long wino_slots(int occ, int cus) {                   // (a multiple of 8 and never 0: a CU-masked stream may report a handful of CUs)
    if (g_wino_slots_override) return g_wino_slots_override;
    const long s = (long)occ * (cus > 0 ? cus : 256) / 8 * 8;
    return s < 8 ? 8 : s;
}

static int wino_pers_enabled() {
    static int v = -1;
    // default OFF: measured 0.4-2 % SLOWER than one tile per workgroup on every Winograd layer of IResNet-50 at B = 128 and 256 (GEMM sum
    // 3 054 against 3 043 us, 5 745 against 5 703: profiles/r05_notes.md) — the hardware's dynamic refill of freed slots is worth more than
    // the hidden first-chunk latency.  1 = on (A / B timing); the test hook fh_debug_wino_slots switches it on for the walk tests.
    if (v < 0) { const char* e = getenv("FACEHIP_WINO_PERS"); v = e ? atoi(e) : 0; }
    return v || g_wino_slots_override;
}

static int wino_mix_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_WINO_MIX"); v = e ? atoi(e) : 1; }          // (0 = uniform F(4x4) tiling everywhere: A / B timing)
    return v;
}

// 128x128 GEMM tiles: 1 = for the mixed layout (default since the stores are whole lines: 14x14x256 76.7 -> 74.5 us at B = 128 — a third
// less L2 -> LDS traffic per FLOP; before that the two tile shapes were level), 2 = everywhere (28x28 and 7x7 lose), 0 = never
static bool wino_bn128(bool mixed) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FACEHIP_WINO_BN128"); v = e ? atoi(e) : 1; }
    return v == 2 || (v == 1 && mixed);
}

// 128x128 or 128x64 tiles for this launch?  The launch runs in whole ROUNDS of resident workgroups (2 per CU for the wide tile, 3 for the
// narrow one) and every tile of a round takes the same time, so the cost of a shape is rounds x tiles-per-CU x work-per-tile — in units of
// one 128x64 tile: 4 per round (wide) against 3 (narrow).  At B = 128 the two are level and the preference above decides (14x14x256: 968 wide
// tiles = 2 rounds = 8 units against 1 936 narrow = 3 rounds = 9); at B = 64 the wide tiles spill 72 of 584 into a second, nearly empty
// round (8 units against 6: 57 us per layer where half of the B = 128 time would be 37).
static bool wino_pick_bn128(long row_tiles, int Cout, bool mixed, int cus) {
    static const int force = [] { const char* e = getenv("FACEHIP_WINO_BN128"); return e ? atoi(e) : -1; }();
    if (force >= 0) return wino_bn128(mixed);
    const long sw = wino_slots(2, cus), sn = wino_slots(3, cus);
    const long wide = (row_tiles * (Cout / 128) + sw - 1) / sw * 4, narrow = (row_tiles * (Cout / 64) + sn - 1) / sn * 3;
    if (wide != narrow) return wide < narrow;
    return wino_bn128(mixed);
}

bool wino_mix_layout(int B, int H, int W, int Cin, int Cout, WinoPlanes* out) {
    const int TY = (H + 3) / 4, TX = (W + 3) / 4;
    const int my = (H & 3) == 1 || (H & 3) == 2, mx = (W & 3) == 1 || (W & 3) == 2;
    // the transform kernel takes one image x 64 channels per workgroup with 16 tiles x 16 float4 columns on its 256 threads; small batches
    // would pad the thin classes' planes (B tiles per plane for the corner class) to mostly empty 128-row GEMM tiles
    if (!wino_mix_enabled() || !(my || mx) || TY * TX != 16 || Cin % 64 || Cout % 64 || B < 64) return false;
    if ((size_t)H * W * 16 * sizeof(v4f) > 64 * 1024) return false;
    WinoPlanes pl{};
    const int n[4] = {(TY - my) * (TX - mx), (TY - my) * mx, my * (TX - mx), my * mx};
    int tile = 0, row = 0, planes = 0;
    for (int c = 0; c < 4; ++c) {
        const int nfr = (c & 2) ? 4 : 6, nfc = (c & 1) ? 4 : 6;
        pl.n[c] = n[c]; pl.nfc[c] = nfc; pl.f2[c] = c;
        pl.rows[c] = (B * n[c] + 127) / 128 * 128;
        pl.t[c] = pl.rows[c] / 128;
        pl.e[c] = tile; pl.base[c] = row;
        tile += nfr * nfc * pl.t[c]; row += nfr * nfc * pl.rows[c];
        if (n[c]) planes += nfr * nfc;
    }
    if ((long)row * std::max(Cin, Cout) * 4 >= (1L << 31)) return false;              // (the transform kernel addresses V / M through 32-bit buffer offsets)
    pl.total_tiles = tile; pl.planes = planes;
    if (out) *out = pl;
    return true;
}

void launch_wino_gemm(const ConvArgs& a, const float* wt36, const float* V, float* M, int cfg, bool bf16x2, hipStream_t s, const WinoPlanes* mix) {
    const int TY = (a.H + 3) / 4, TX = (a.W + 3) / 4;
    const long NT = (long)a.B * TY * TX;
    if (NT <= 0) return;
    const long NTp = wino_rows(NT);                          // rows per frequency plane, padded to whole GEMM tiles
    KernelTimer& timer = KernelTimer::get();
    ConvArgs g{};
    g.in = V; g.wt = wt36; g.out1 = M; g.slabs = a.slabs; g.sk_err = a.sk_err; g.sk_enable = a.sk_enable; g.cus = a.cus;
    g.B = (int)(36 * NTp); g.H = g.W = g.Ho = g.Wo = 1; g.Cin = a.Cin; g.Cout = a.Cout; g.ks = 1; g.stride = 1; g.pad = 0; g.Kpad = a.Cin;
    g.act = (int)Act::NONE; g.res_mode = (int)ResMode::NONE;
    g.wt_group_rows = (int)NTp; g.wt_gs = (long)conv_wt_rows(a.Cout) * a.Cin;
    // booked on the GEMM: the FLOPs it EXECUTES on real rows (physical matrix-core utilisation) and, in the bytes slot of the
    // timer, the direct-form FLOPs of the convolution it stands for (the algorithmic figure bench.py quotes beside it)
    g.t_flops = 2.0 * 36.0 * (double)NT * a.Cin * a.Cout; g.t_bytes = a.t_flops;
    WinoPlanes pl{};
    if (mix) {
        if (!(a.Cout % 32 == 0 && cfg == 2)) throw std::runtime_error("winograd: the mixed tiling runs on the lean GEMM kernel only");
        pl = *mix;
        double real_rows = 0;
        for (int c = 0; c < 4; ++c) real_rows += (double)((c & 2) ? 4 : 6) * pl.nfc[c] * a.B * pl.n[c];
        g.t_flops = 2.0 * real_rows * a.Cin * a.Cout;
    } else {
        pl.e[0] = 0; pl.e[1] = pl.e[2] = pl.e[3] = (int)(36 * NTp / 128); pl.t[0] = (int)(NTp / 128); pl.nfc[0] = 6;
        pl.total_tiles = pl.e[1];
    }
    if (a.Cout % 32 == 0 && cfg == 2) {
        // the lean kernel: 128 x 64 tiles (IResNet-50 at B = 128 / 256: 9.89 / 19.5 ms; 128 x 32: 9.96 / 19.8; 128 x 128: 10.2 / 19.6)
        const long rows = 128L * pl.total_tiles;
        const int chunks = a.Cin / 32;
        const bool wide = a.Cout % 64 == 0;
        timer.begin(s);
        if (bf16x2) {
            if (!wino_gemm_ok_bf16x2(a.Cin, a.Cout)) throw std::runtime_error("winograd: this layer has no split-bf16 GEMM");
            hipLaunchKernelGGL((wino_gemm_bf16x2_kernel<64, 3>), dim3((unsigned)((rows / 128) * (a.Cout / 64))), dim3(256), 0, s, V, wt36, M, a.Cin,
                               a.Cout, pl, g.wt_gs, a.Cout / 64, chunks);
        } else if (wide && a.Cout % 128 == 0 && wino_pick_bn128(rows / 128, a.Cout, mix != nullptr, a.cus)) {
            const long nt = (rows / 128) * (a.Cout / 128), slots = wino_slots(2, a.cus);
            if (wino_pers_enabled() && nt > slots)
                hipLaunchKernelGGL((wino_gemm_pers_kernel<128, 2>), dim3((unsigned)slots), dim3(256), 0, s, V, wt36, M, a.Cin, a.Cout, pl, g.wt_gs,
                                   a.Cout / 128, chunks, (int)nt);
            else
                hipLaunchKernelGGL((wino_gemm_kernel<128, 2>), dim3((unsigned)nt), dim3(256), 0, s, V, wt36, M, a.Cin, a.Cout, pl, g.wt_gs, a.Cout / 128,
                                   chunks);
        } else if (wide && wino_pers_enabled() && (rows / 128) * (a.Cout / 64) > wino_slots(3, a.cus)) {
            const long nt = (rows / 128) * (a.Cout / 64), slots = wino_slots(3, a.cus);
            hipLaunchKernelGGL((wino_gemm_pers_kernel<64, 3>), dim3((unsigned)slots), dim3(256), 0, s, V, wt36, M, a.Cin, a.Cout, pl, g.wt_gs, a.Cout / 64,
                               chunks, (int)nt);
        } else if (wide) {
#ifdef FACEHIP_WINO_ABL
            static const int abl = [] { const char* e = getenv("FACEHIP_WINO_ABL"); return e ? atoi(e) : 0; }();
            const dim3 grid((unsigned)((rows / 128) * (a.Cout / 64)));
#define WABL(X) case X: hipLaunchKernelGGL((wino_gemm_kernel<64, 3, X>), grid, dim3(256), 0, s, V, wt36, M, a.Cin, a.Cout, pl, g.wt_gs, a.Cout / 64, chunks); break;
            switch (abl) { WABL(1) WABL(2) WABL(3) WABL(4) WABL(5) WABL(6) WABL(7) WABL(8) WABL(15) WABL(16) WABL(17) WABL(24)
                default: hipLaunchKernelGGL((wino_gemm_kernel<64, 3>), grid, dim3(256), 0, s, V, wt36, M, a.Cin, a.Cout, pl, g.wt_gs, a.Cout / 64, chunks); }
#undef WABL
#else
            hipLaunchKernelGGL((wino_gemm_kernel<64, 3>), dim3((unsigned)((rows / 128) * (a.Cout / 64))), dim3(256), 0, s, V, wt36, M, a.Cin, a.Cout,
                                   pl, g.wt_gs, a.Cout / 64, chunks);
#endif
        }
        else
            hipLaunchKernelGGL((wino_gemm_kernel<32, 4>), dim3((unsigned)((rows / 128) * (a.Cout / 32))), dim3(256), 0, s, V, wt36, M, a.Cin, a.Cout,
                               pl, g.wt_gs, a.Cout / 32, chunks);
        timer.end(s, 7, g.t_flops, g.t_bytes);
    } else {
        if (bf16x2) throw std::runtime_error("winograd: this layer has no split-bf16 GEMM");
        launch_conv(g, cfg, s);                              // generic grouped instantiation of conv_igemm_kernel
    }
}

static WinoOutArgs wino_out_args(const ConvArgs& a, const float* M) {
    const int TY = (a.H + 3) / 4, TX = (a.W + 3) / 4;
    WinoOutArgs o{};
    o.M = M; o.bias = a.bias; o.slope = a.slope; o.res = a.res; o.out1 = a.out1; o.out2 = a.out2; o.s2 = a.s2; o.t2 = a.t2;
    o.B = a.B; o.H = a.H; o.W = a.W; o.C = a.Cout; o.TY = TY; o.TX = TX; o.act = a.act; o.NTp = wino_rows((long)a.B * TY * TX);
    return o;
}

// stage 3: M -> out (bias -> activation -> + residual -> out1, optional out2 = out1 * s2 + t2)
void launch_wino_output(const ConvArgs& a, const float* M, hipStream_t s) {
    const WinoOutArgs o = wino_out_args(a, M);
    const long NT = (long)a.B * o.TY * o.TX;
    if (NT <= 0) return;
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL(wino_output_kernel, dim3(wino_grid(NT * (a.Cout >> 2))), dim3(256), 0, s, o);
    timer.end(s, 8, 0.0, 0.0);
}

// can stage 3 of this convolution be fused with stage 1 of a following Winograd convolution on the same map?
static bool wino_fuse_shape(int H, int W, int C, int* csl4, int* pitch, int* img) {
    const int tpi = ((H + 3) / 4) * ((W + 3) / 4);
    int c = 0, pt = 0, im = 0;
    if (C % 64 == 0 && tpi <= 16 && (tpi & (tpi - 1)) == 0) { c = 16; pt = 16; im = 16 / tpi; }
    else if (C % 16 == 0 && tpi <= 64) { c = 4; pt = 5; im = 64 / tpi; }
    else return false;
    if ((size_t)im * H * W * pt * sizeof(v4f) > 64 * 1024) return false;     // the LDS image of `im` pictures
    if (csl4) { *csl4 = c; *pitch = pt; *img = im; }
    return true;
}
// touches_memory: the fused kernel would also read a residual or write out1 / out2.  On maps of more than 16 tiles it works on 16-channel
// slices, i.e. 64-byte pieces of every pixel row: fine for the M / V planes, but it halves the efficiency of those activation accesses
// (28x28x128, B = 128: 116 us fused against 60 + 46 us separate) — such junctions keep the two separate kernels.
bool wino_can_fuse(int H, int W, int C, bool touches_memory) {
    if (wino_slice_shape(H, W, C)) return true;                // 32-channel slices: whole lines either way
    int csl4 = 0, pitch = 0, img = 0;
    if (!wino_fuse_shape(H, W, C, &csl4, &pitch, &img)) return false;
    return csl4 == 16 || !touches_memory;
}

// stage 3 of convolution `a` + stage 1 of the next one in one kernel: M -> (out1 / out2 if non-null) and -> V of the next convolution,
// which sees out1 (feed_aff = 0) or out1 * s2 + t2 (feed_aff = 1; a.s2 / a.t2 must then be set even when a.out2 is null).
void launch_wino_fused(const ConvArgs& a, const float* M, float* Vnext, int feed_aff, bool pack_next, hipStream_t s) {
    WinoFuseArgs f{};
    if (wino_slice_shape(a.H, a.W, a.Cout)) {
        f.o = wino_out_args(a, M);
        f.V = Vnext; f.feed_aff = feed_aff; f.pack = pack_next ? 1 : 0; f.csl4 = 8; f.pitch = 8; f.img = wino_slice_enabled() == 2 ? 2 : 1;
        if (a.B <= 0) return;
        wino_check((long)a.B * f.o.TY * f.o.TX, a.Cout);
        const size_t lds = (size_t)a.H * 32 * 8 * sizeof(v4f);
        static const bool attr = [] {                          // > 64 KB of dynamic LDS must be asked for once
            FH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_slice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return true;
        }();
        (void)attr;
        KernelTimer& timer = KernelTimer::get();
        timer.begin(s);
        hipLaunchKernelGGL(wino_slice_kernel, dim3((unsigned)(a.B * (a.Cout / 32))), dim3(kSliceThreads), lds, s, f);
        timer.end(s, 8, 0.0, 0.0);
        return;
    }
    if (!wino_fuse_shape(a.H, a.W, a.Cout, &f.csl4, &f.pitch, &f.img)) throw std::runtime_error("winograd: this map cannot take the fused transform");
    f.o = wino_out_args(a, M);
    f.V = Vnext; f.feed_aff = feed_aff; f.pack = pack_next ? 1 : 0;
    const long NT = (long)a.B * f.o.TY * f.o.TX;
    if (NT <= 0) return;
    wino_check(NT, a.Cout);
    const size_t lds = (size_t)f.img * a.H * a.W * f.pitch * sizeof(v4f);
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL(wino_fused_kernel, dim3((unsigned)(((a.B + f.img - 1) / f.img) * (a.Cout / (4 * f.csl4)))), dim3(256), lds, s, f);
    timer.end(s, 8, 0.0, 0.0);
}

// ---- mixed F(4x4) / F(2x2) tiling: the transform kernel (see WinoPlanes in kernels.h) --------------------------------------------------
// wino_fused_kernel's anatomy (one image x 64 channels per workgroup, thread = (tile, float4 column), LDS image between the two phases)
// with the tile's class decided per thread: the row direction of a tile is wave-uniform (wave = tile row), the column direction differs
// between the 16-lane groups of a wave.  Both 1-D transforms are evaluated and selected (the kernel is bound by its V / M traffic, not by
// the vector ALU); frequencies a class does not have are addressed OUT OF RANGE of the buffer descriptor, so their loads return zero
// and their stores are dropped by the hardware — no branch, and the 36 loads of a thread stay in flight together.
// F(2x2,3x3) with the points {0, 1, -1, inf}: B^T rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3), A^T = [1 1 1 0; 0 1 -1 -1]; its G rows are
// (4, -3, -3, 1) x the rows {0, 1, 2, 5} of F(4x4,3x3)'s G, so with those factors on the B^T rows the planes multiply the F(4x4) weight
// matrices U[6 i + j] unchanged.
struct WinoMixArgs {
    WinoOutArgs o;          // (o.M null: the image comes from `in`)
    WinoPlanes pl;
    const float* in; const float* in_s; const float* in_t;
    float* V;
    unsigned plane_bytes;   // size of the V / M workspace in this layout (bytes): the buffer descriptors' range
    int feed_aff, pack;
};

__device__ __forceinline__ void wino_bt2s(const v4f (&d)[6], v4f (&t)[6]) {
    t[0] = 4.f * (d[0] - d[2]);
    t[1] = -3.f * (d[1] + d[2]);
    t[2] = 3.f * (d[1] - d[2]);
    t[3] = d[1] - d[3];
    t[4] = v4f{0.f, 0.f, 0.f, 0.f};
    t[5] = t[4];
}

__global__ __launch_bounds__(256, 2) void wino_mix_kernel(const WinoMixArgs a) {
    const WinoOutArgs& p = a.o;
    extern __shared__ v4f act[];                               // [H*W][16 float4]
    const int C = p.C, HW = p.H * p.W;
    const int tid = threadIdx.x, c4l = tid & 15, tile = tid >> 4;
    const int cslices = C >> 6;
    const int cs = blockIdx.x % cslices, b = blockIdx.x / cslices;
    const int ty = tile / p.TX, tx = tile - ty * p.TX;
    const int my = a.pl.n[2] + a.pl.n[3] > 0, mx = a.pl.n[1] + a.pl.n[3] > 0;
    const bool rF2 = my && ty == p.TY - 1, cF2 = mx && tx == p.TX - 1;
    const int nfr = rF2 ? 4 : 6, nfc = cF2 ? 4 : 6;
    // row of this tile in plane 0 of its class, and the class's plane stride (constant indices: a lane-dependent index into the argument
    // struct would go through scratch memory)
    const int n_c = rF2 ? (cF2 ? a.pl.n[3] : a.pl.n[2]) : (cF2 ? a.pl.n[1] : a.pl.n[0]);
    const int rows_c = rF2 ? (cF2 ? a.pl.rows[3] : a.pl.rows[2]) : (cF2 ? a.pl.rows[1] : a.pl.rows[0]);
    const int base_c = rF2 ? (cF2 ? a.pl.base[3] : a.pl.base[2]) : (cF2 ? a.pl.base[1] : a.pl.base[0]);
    const int lidx = rF2 ? (cF2 ? 0 : tx) : (cF2 ? ty : ty * (p.TX - mx) + tx);
    const int c4 = cs * 16 + c4l;
    const unsigned off0 = (unsigned)(((base_c + b * n_c + lidx) * C + c4 * 4) * 4);       // byte offset of frequency (0, 0)
    const unsigned pstride = (unsigned)rows_c * (unsigned)C * 4u;                            // bytes between frequency planes
    constexpr unsigned OOB = 0x80000000u;
    if (p.M) {
        // ---- phase 1: Y = A^T M A, epilogue, LDS image
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.M), 0, a.plane_bytes, 0x00020000);
        v4f t[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            v4f m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool ok = i < nfr && j < nfc;
                m[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rm, ok ? off0 + (unsigned)(i * nfc + j) * pstride : OOB, 0, 0));
            }
            v4f y4[4];
            wino_at(m, y4);
            const v4f y20 = m[0] + m[1] + m[2], y21 = m[1] - m[2] - m[3];
            t[0][j] = rF2 ? y20 : y4[0];
            t[1][j] = rF2 ? y21 : y4[1];
            t[2][j] = y4[2];                                  // (rows 2, 3 of an F(2) tile lie outside the map: never stored)
            t[3][j] = y4[3];
        }
        const v4f b4 = p.bias ? *reinterpret_cast<const v4f*>(p.bias + c4 * 4) : v4f{0.f, 0.f, 0.f, 0.f};
        v4f rs[4][4];                                        // residual reads before the first store (see wino_output_kernel)
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int oy = 4 * ty + r, ox = 4 * tx + x;
                    rs[r][x] = *reinterpret_cast<const v4f*>(p.res + (((size_t)b * p.H + min(oy, p.H - 1)) * p.W + min(ox, p.W - 1)) * C + c4 * 4);   // (clamped: unconditional)
                }
        }
        v4f sl = {0.f, 0.f, 0.f, 0.f}, s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
        if (p.act == (int)Act::PRELU) sl = *reinterpret_cast<const v4f*>(p.slope + c4 * 4);
        if (p.s2) { s2 = *reinterpret_cast<const v4f*>(p.s2 + c4 * 4); t2 = *reinterpret_cast<const v4f*>(p.t2 + c4 * 4); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 4 * ty + r;
            if (oy >= p.H) continue;
            v4f y[4];
            wino_at(t[r], y);
            if (cF2) { y[0] = t[r][0] + t[r][1] + t[r][2]; y[1] = t[r][1] - t[r][2] - t[r][3]; }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int ox = 4 * tx + x;
                if (ox >= p.W) continue;
                v4f v = y[x] + b4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float u = v[e];
                    if (p.act == (int)Act::RELU) u = u > 0.f ? u : 0.f;
                    else if (p.act == (int)Act::PRELU) u = u >= 0.f ? u : u * sl[e];
                    else if (p.act == (int)Act::SIGMOID) u = 1.0f / (1.0f + expf(-u));
                    v[e] = u;
                }
                const size_t o = (((size_t)b * p.H + oy) * p.W + ox) * C + c4 * 4;
                if (p.res) v += rs[r][x];
                if (p.out1) *reinterpret_cast<v4f*>(p.out1 + o) = v;
                const v4f vb = v * s2 + t2;
                if (p.out2) *reinterpret_cast<v4f*>(p.out2 + o) = vb;
                act[(oy * p.W + ox) * 16 + c4l] = a.feed_aff ? vb : v;
            }
        }
    } else {
        // ---- the image comes from memory (first convolution of a chain): whole 256-byte pixel rows per 16 lanes
        v4f as4 = {1.f, 1.f, 1.f, 1.f}, at4 = {0.f, 0.f, 0.f, 0.f};
        if (a.in_s) { as4 = *reinterpret_cast<const v4f*>(a.in_s + c4 * 4); at4 = *reinterpret_cast<const v4f*>(a.in_t + c4 * 4); }
        const float* src = a.in + (size_t)b * HW * C + c4 * 4;
        for (int px = tile; px < HW; px += 16) act[px * 16 + c4l] = *reinterpret_cast<const v4f*>(src + (size_t)px * C) * as4 + at4;
    }
    if (!a.V) return;
    __syncthreads();
    // ---- phase 2: V = B^T d B from the LDS image (zeros outside the map)
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(a.V, 0, a.plane_bytes, 0x00020000);
    const v4f* img = act + c4l;
    const int iy0 = 4 * ty - 1, ix0 = 4 * tx - 1;
    v4f tt[6][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int ix = ix0 + c;
        v4f d[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = iy0 + r;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            d[r] = ok ? img[(iy * p.W + ix) * 16] : v4f{0.f, 0.f, 0.f, 0.f};
        }
        v4f t4[6], t2s[6];
        wino_bt(d, t4);
        wino_bt2s(d, t2s);
#pragma unroll
        for (int i = 0; i < 6; ++i) tt[i][c] = rF2 ? t2s[i] : t4[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        v4f o4[6], o2[6];
        wino_bt(tt[i], o4);
        wino_bt2s(tt[i], o2);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bool ok = i < nfr && j < nfc;
            const v4f o = cF2 ? o2[j] : o4[j];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, a.pack ? wino_pack_bf16x2(o) : o), rv,
                                                   ok ? off0 + (unsigned)(i * nfc + j) * pstride : OOB, 0, 0);
        }
    }
}

void launch_wino_mix(const ConvArgs& a, const WinoPlanes& pl, const float* M, float* Vnext, int feed_aff, const float* in_scale,
                     const float* in_shift, bool pack_next, hipStream_t s) {
    if (a.B <= 0) return;
    WinoMixArgs f{};
    f.o = wino_out_args(a, M);
    if (!M) { f.o.C = a.Cin; f.o.res = nullptr; f.o.out1 = f.o.out2 = nullptr; }
    f.pl = pl;
    f.in = a.in; f.in_s = in_scale; f.in_t = in_shift;
    f.V = Vnext; f.feed_aff = feed_aff; f.pack = pack_next ? 1 : 0;
    long rows = 0;
    for (int c = 0; c < 4; ++c) rows += (long)((c & 2) ? 4 : 6) * pl.nfc[c] * pl.rows[c];
    f.plane_bytes = (unsigned)(rows * f.o.C * 4);
    if (f.o.C % 64 || f.o.TY * f.o.TX != 16 || rows * f.o.C * 4 >= (1L << 31)) throw std::runtime_error("winograd: this map cannot take the mixed tiling");
    const size_t lds = (size_t)a.H * a.W * 16 * sizeof(v4f);
    KernelTimer& timer = KernelTimer::get();
    timer.begin(s);
    hipLaunchKernelGGL(wino_mix_kernel, dim3((unsigned)(a.B * (f.o.C / 64))), dim3(256), lds, s, f);
    timer.end(s, 8, 0.0, 0.0);
}

// all three stages of one convolution (used by the single-layer test entry point)
void launch_conv_winograd(const ConvArgs& a, const float* wt36, float* V, float* M, int cfg, const float* in_scale, const float* in_shift,
                          hipStream_t s) {
    WinoPlanes pl;
    if (cfg == 2 && wino_mix_layout(a.B, a.H, a.W, a.Cin, a.Cout, &pl)) {
        launch_wino_mix(a, pl, nullptr, V, 0, in_scale, in_shift, false, s);
        launch_wino_gemm(a, wt36, V, M, cfg, false, s, &pl);
        launch_wino_mix(a, pl, M, nullptr, 0, nullptr, nullptr, false, s);
        return;
    }
    launch_wino_input(a, V, in_scale, in_shift, false, s);
    launch_wino_gemm(a, wt36, V, M, cfg, false, s);
    launch_wino_output(a, M, s);
}

}  // namespace fh
