// gallery.hip — 1:N generalisation of FaceRecognizer::compareFaces (reference src/face_recognizer.cpp:320-334): every query against
// every enrolled row, mapped score (dot + 1) / 2, top-k per query ranked (score desc, global row index asc).
//
// One kernel streams the gallery ONCE: the dot products live only in MFMA accumulators, never in memory.
//   * GEMM part = the tile anatomy of the other f32 kernels (LDS-DMA global_load_lds_dwordx4, [row][32 k] LDS images with the 16-byte
//     column XOR-swizzled on the source side, one ds_read_b128 per 4 v_mfma_f32_32x32x2_f32).  Workgroup tile: 128 gallery rows x 64
//     queries, K = dim in 32-deep chunks; the MFMA's A operand is the GALLERY fragment and B the QUERY fragment, so a lane ends up with
//     ONE query (column) and 16 gallery rows of it per 32x32 block.
//   * top-k part: every workgroup owns a contiguous run of row tiles and keeps, per query, a sorted k-list and its k-th entry (the
//     admission threshold) in LDS.  After a tile's K loop each lane compares its 32 scores with the threshold of its query; the few that
//     pass are appended to that query's slot queue (LDS atomic counter) and 64 threads — one per query — insert them.  A queue holds 32
//     entries: if a tile overflows one (only the first tiles of a run can, while the lists are still filling), the tile's scores — still
//     in registers — are replayed in four 32-row rounds, which cannot overflow.
//   * per-workgroup lists go to memory as [part][Q][k]; topk_merge_kernel (face_kernels.hip) selects the overall top-k.
// Bounds: HBM scan (G x dim x 4 bytes once) against 2*Q*G*dim FLOP on the f32 matrix cores — at Q = 64 the two meet.
#include <hip/hip_runtime.h>

#include <climits>
#include <stdexcept>

#include "kernels.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void gal_dma16(const float* src, v4f* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
    (void)src; (void)dst;
#endif
}

constexpr int GAL_BM = 128, GAL_BN = 64, GAL_KMAX = 16, GAL_QCAP = 32;

struct GalArgs {
    const float* gal;       // [G][dim]
    const float* q;         // [tiles_n * 64][dim], rows >= Q are zero
    const float* zeros;
    long G, idx_base;
    int dim, Q, k, tiles_n, row_tiles, tiles_per_part;
    float* ps;              // [parts][Q][k]
    int* pi;
};

__device__ __forceinline__ bool gal_better(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

__global__ __launch_bounds__(256, 2) void gallery_topk_kernel(const GalArgs p) {
    constexpr int BM = GAL_BM, BN = GAL_BN, TN = BN / 32, AL = BM / 32, BL = BN / 32;
    __shared__ v4f lds[2][(BM + BN) * 8];
    __shared__ float lst_s[BN][GAL_KMAX];
    __shared__ int lst_i[BN][GAL_KMAX];
    __shared__ float que_s[BN][GAL_QCAP];
    __shared__ int que_i[BN][GAL_QCAP];
    __shared__ int cnt[BN];
    __shared__ int overflow;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;
    // XCD-contiguous block order: the tiles_n workgroups that stream the same rows sit on one XCD (one L2)
    int t;
    {
        const int nb = gridDim.x, qq = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        t = x * qq + min(x, r8) + (int)(blockIdx.x >> 3);
    }
    const int tile_n = t % p.tiles_n, part = t / p.tiles_n;
    const int n0 = tile_n * BN;
    const int K = p.dim, chunks = K / 32, k = p.k;
    const int rt0 = part * p.tiles_per_part, rt1 = min(p.row_tiles, rt0 + p.tiles_per_part);

    for (int i = tid; i < BN * GAL_KMAX; i += 256) { (&lst_s[0][0])[i] = -1.0f; (&lst_i[0][0])[i] = INT_MAX; }
    if (tid < BN) cnt[tid] = 0;
    if (tid == 0) overflow = 0;

    const int lrow = tid >> 3;
    const int lqs = (tid & 7) ^ ((lrow >> 1) & 7);
    const size_t row32 = (size_t)32 * K;
    v4f* const dstA = &lds[0][wid * 64];
    v4f* const dstB = &lds[0][BM * 8 + wid * 64];
    const float* const b_base = p.q + (size_t)(n0 + lrow) * K + lqs * 4;

    for (int rt = rt0; rt < rt1; ++rt) {
        const long m0 = (long)rt * BM;
        const float* a_src[AL];
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const long r = m0 + lrow + 32 * i;
            a_src[i] = r < p.G ? p.gal + (size_t)r * K + lqs * 4 : p.zeros;     // rows past the end: zero line (and masked below)
        }
        const float* b_src = b_base;
        const bool tail = m0 + BM > p.G;
        auto load_chunk = [&](int buf) {
            v4f* const dA = dstA + buf * ((BM + BN) * 8);
            v4f* const dB = dstB + buf * ((BM + BN) * 8);
#pragma unroll
            for (int i = 0; i < AL; ++i) { gal_dma16(a_src[i], dA + i * 32 * 8); if (!tail || m0 + lrow + 32 * i < p.G) a_src[i] += 32; }
#pragma unroll
            for (int i = 0; i < BL; ++i) gal_dma16(b_src + i * row32, dB + i * 32 * 8);
            b_src += 32;
        };
        v16f acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        __syncthreads();                                   // previous tile's epilogue is done with everything
        load_chunk(0);
        __syncthreads();
        for (int kc = 0; kc < chunks; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < chunks) load_chunk(buf ^ 1);
            const v4f* X = lds[buf] + (wid * 32 + fr) * 8;         // gallery rows of this wave (MFMA A operand)
            const v4f* Wt = lds[buf] + BM * 8 + fr * 8;            // queries (MFMA B operand)
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
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[e], w[j][e], acc[j], 0, 0, 0);
            }
            __syncthreads();
        }
        // ---- top-k epilogue.  C/D map: column (= query) = lane & 31, row (= gallery row of the wave's 32) = (e&3) + 8*(e>>2) + 4*(lane>>5)
        const long rbase = m0 + wid * 32 + 4 * fh2;
        auto push = [&](int g_lo, int g_hi) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int qi = j * 32 + fr;
                const float ts = lst_s[qi][k - 1];
                const int ti = lst_i[qi][k - 1];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if ((e >> 2) < g_lo || (e >> 2) >= g_hi) continue;
                    const long row = rbase + 8 * (e >> 2) + (e & 3);
                    const float sc = (acc[j][e] + 1.0f) / 2.0f;          // the reference's mapping (face_recognizer.cpp:331-333)
                    const int gi = (int)(p.idx_base + row);
                    if (row < p.G && gal_better(sc, gi, ts, ti)) {
                        const int slot = atomicAdd(&cnt[qi], 1);
                        if (slot < GAL_QCAP) { que_s[qi][slot] = sc; que_i[qi][slot] = gi; }
                        else overflow = 1;
                    }
                }
            }
        };
        auto insert = [&]() {                                // one thread per query: its queue into its sorted list
            if (tid < BN) {
                const int n = min(cnt[tid], GAL_QCAP);
                for (int c = 0; c < n; ++c) {
                    const float s = que_s[tid][c];
                    const int gi = que_i[tid][c];
                    if (!gal_better(s, gi, lst_s[tid][k - 1], lst_i[tid][k - 1])) continue;
                    int pos = k - 1;
                    while (pos > 0 && gal_better(s, gi, lst_s[tid][pos - 1], lst_i[tid][pos - 1])) {
                        lst_s[tid][pos] = lst_s[tid][pos - 1]; lst_i[tid][pos] = lst_i[tid][pos - 1];
                        --pos;
                    }
                    lst_s[tid][pos] = s; lst_i[tid][pos] = gi;
                }
                cnt[tid] = 0;
            }
        };
        push(0, 4);
        __syncthreads();
        if (overflow) {                                      // (block-uniform: read after the barrier) replay in 32-row rounds
            __syncthreads();
            if (tid < BN) cnt[tid] = 0;
            if (tid == 0) overflow = 0;
            __syncthreads();
            for (int g = 0; g < 4; ++g) {
                push(g, g + 1);
                __syncthreads();
                insert();
                __syncthreads();
            }
        } else {
            insert();
        }
    }
    __syncthreads();
    for (int i = tid; i < BN * k; i += 256) {
        const int qi = i / k, pos = i - qi * k, qg = n0 + qi;
        if (qg >= p.Q) continue;
        const int gi = lst_i[qi][pos];
        const size_t o = ((size_t)part * p.Q + qg) * k + pos;
        p.ps[o] = gi == INT_MAX ? -1.0f : lst_s[qi][pos];
        p.pi[o] = gi == INT_MAX ? -1 : gi;
    }
}

// parts the row range is cut into for a gallery of G rows and a query batch of Q (the caller sizes its partial-list buffers with it)
int gallery_parts(long G, int Q, int* tiles_per_part) {
    const int tiles_n = (Q + GAL_BN - 1) / GAL_BN;
    const long row_tiles = (G + GAL_BM - 1) / GAL_BM;
    const int slots = conv_num_cus() * 2;                       // 2 resident workgroups per CU
    long parts = slots / tiles_n;
    if (parts < 1) parts = 1;
    if (parts > row_tiles) parts = row_tiles;
    const long tpp = parts > 0 ? (row_tiles + parts - 1) / parts : 1;
    if (tiles_per_part) *tiles_per_part = (int)tpp;
    return (int)(tpp > 0 ? (row_tiles + tpp - 1) / tpp : 0);
}

// queries: packed [ceil64(Q)][dim] with zero rows behind Q; part_score / part_idx: [gallery_parts][Q][k]
void launch_gallery_topk(const float* gal, long G, int dim, const float* qpacked, int Q, int k, long idx_base, float* part_score, int* part_idx,
                         hipStream_t s) {
    if (G <= 0 || Q <= 0) return;
    if (dim % 32 || k < 1 || k > GAL_KMAX) throw std::runtime_error("gallery: need dim % 32 == 0 and 1 <= k <= 16");
    if (idx_base + G > (long)INT_MAX) throw std::runtime_error("gallery: global row indices must fit in 31 bits");
    GalArgs a{};
    a.gal = gal; a.q = qpacked; a.zeros = conv_zero_line(); a.G = G; a.idx_base = idx_base; a.dim = dim; a.Q = Q; a.k = k;
    a.tiles_n = (Q + GAL_BN - 1) / GAL_BN;
    a.row_tiles = (int)((G + GAL_BM - 1) / GAL_BM);
    const int parts = gallery_parts(G, Q, &a.tiles_per_part);
    a.ps = part_score; a.pi = part_idx;
    hipLaunchKernelGGL(gallery_topk_kernel, dim3((unsigned)(parts * a.tiles_n)), dim3(256), 0, s, a);
}

}  // namespace fh
