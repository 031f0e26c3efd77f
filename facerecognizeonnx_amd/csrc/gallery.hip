// gallery.hip — 1:N generalisation of FaceRecognizer::compareFaces (reference src/face_recognizer.cpp:320-334): every query against
// every enrolled row, mapped score (dot + 1) / 2, top-k per query ranked (score desc, global row index asc).
//
// One kernel streams the gallery ONCE: the dot products live only in MFMA accumulators, never in memory.
//   * The scan is an HBM stream with NO reuse on the gallery side: what limits it is bytes in flight (Little: ~50 KB per CU for 5 TB/s at
//     ~2.5 us loaded latency).  Each wave fetches ITS OWN 32 gallery rows straight into registers — the fragment layout is the one a
//     ds_read_b128 would deliver: lane (row fr, half fh2) takes 16 bytes at k = (2s + fh2) * 4 of its row; the s-steps of a 128-byte line
//     are consecutive instructions — in 64-deep chunks, one chunk (8 loads, 32 VGPRs) ahead of the one being multiplied: 8 waves x 8 KB in
//     flight per CU, no LDS traffic and no barrier on the gallery side.  Only the 64 queries of the tile (shared by the four waves) go
//     through LDS: 16 KB per chunk by LDS-DMA from L2, double buffered, 16-byte column XOR-swizzled by (row & 15) on the source side.
//   * v_mfma_f32_32x32x2_f32 with the GALLERY fragment as A and the QUERY fragment as B: a lane ends up with ONE query (column) and 16
//     gallery rows of it per 32x32 block.  Workgroup tile: 128 gallery rows x 64 queries.
//   * top-k: every workgroup owns a contiguous run of row tiles.  Thread q < 64 keeps query q's sorted k-list in REGISTERS (a compare-
//     exchange pass per insertion, no LDS latency chain) and publishes its k-th entry — the admission threshold — in LDS.  After a tile's
//     K loop each lane compares its 32 scores with the threshold of its query; the few that pass are appended to that query's slot queue
//     (LDS atomic counter) and thread q inserts them.  A queue holds 32 entries: if a tile overflows one (only the first tiles of a run
//     can, while the lists are still filling), the tile's scores — still in registers — are replayed in four 32-row rounds, which cannot.
//   * per-workgroup lists go to memory as [part][Q][k]; topk_merge_kernel (face_kernels.hip) selects the overall top-k.
// Bounds: HBM scan (G x dim x 4 bytes once) against 2*Q*G*dim FLOP on the f32 matrix cores — at Q = 64 the two meet.
//
// Round 3, measured and NOT kept (1 M x 512, Q = 64, k = 16; this kernel: 0.82 ms per call = seed pass 88 us + scan ~700 + two merges
// 39 us each): a scan with the whole 64-query tile RESIDENT in LDS (128 KB, one 8-wave workgroup per CU, no per-chunk barrier, query
// fragments read a step ahead), tried with three top-k schemes:
//   * this kernel's queues + two barriers per 256-row tile + its own seed launch: 0.86 ms (the seed launch alone 196 us: sixteen
//     workgroups each loading 128 KB of queries for one tile; a barrier stalls the whole CU on its slowest wave);
//   * wave-private lists in registers (lane l = query l, one lane^32 exchange per accumulator position, one compare-exchange pass per
//     candidate), thresholds shared through LDS, no barrier and no seed: 0.91 ms — the K loop + loads alone 0.67 ms, but the
//     insertions cost 0.3 ms: the k-th score of ONE wave's rows (or the maximum of several waves' k-ths) is a far weaker threshold
//     than the k-th of their union, so ~20 of the 32 insertion passes of a tile still fire half-way through the scan;
//   * the same with chip-wide thresholds through atomicMax on 64 words: 1.18 ms (contended atomics, and the maximum of per-workgroup
//     k-ths is still not a chip-wide k-th).
// Where a call's 0.82 ms go (rocprofv3, 1 M x 512, Q = 64, k = 16): seed scan 89 us (32 workgroups, one tile each: a latency chain plus the
// first-tile insertions) + seed merge 26 + main scan 658 (67.1 GFLOP = 102 TFLOP/s: at Q = 64 the f32 matrix cores bind, not HBM —
// 0.43 ms at the nominal peak, ~0.58 at the 115 TFLOP/s plateau) + final merge 50.  Without the seed pass (FACEHIP_GAL_SEED=0) the call
// takes 0.844 ms: the per-workgroup list warm-up costs more than the 115 us the seed does.  A pruned final merge (threshold = best over
// the parts of a full part's worst entry, survivors compacted to LDS, one entry per thread in the rounds) ran 35 us SLOWER with
// part-per-thread loads (64 lines per load instruction) and was not kept.
// Side results worth keeping: a wave streaming its own 32 rows into registers reaches 6.2 TB/s whatever the lane-to-row mapping
// (scripts/ubench/row_stream.hip: 32, 16, 8 rows per instruction or fully coalesced, all 6.2-6.5 TB/s), so the scan is not bound by
// its access pattern; with loads and top-k switched off the MFMA + fragment-read loop alone runs at ~75 % of the f32 peak.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>
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
    const float* seed_s;    // optional [Q][k]: exact top-k of a PREFIX of the gallery — its k-th entry is a valid admission threshold for
    const int* seed_i;      // the whole scan (k rows at least as good exist), so the per-workgroup lists start almost closed
};

__device__ __forceinline__ bool gal_better(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

__global__ __launch_bounds__(256, 2) void gallery_topk_kernel(const GalArgs p) {
    constexpr int BM = GAL_BM, BN = GAL_BN, TN = BN / 32;
    __shared__ v4f ldsq[2][BN * 16];                          // query chunk [64 rows][64 k], 16-byte column XOR (row & 15)
    __shared__ float tau_s[BN];                               // admission threshold per query = the k-th entry of its list
    __shared__ int tau_i[BN];
    __shared__ float que_s[BN][GAL_QCAP];
    __shared__ int que_i[BN][GAL_QCAP];
    __shared__ int cnt[BN];
    __shared__ int overflow;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh2 = lane >> 5;
    int t;
    {
        const int nb = gridDim.x, qq = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        t = x * qq + min(x, r8) + (int)(blockIdx.x >> 3);
    }
    const int tile_n = t % p.tiles_n, part = t / p.tiles_n;
    const int n0 = tile_n * BN;
    const int K = p.dim, chunks = K / 64, k = p.k;
    const int rt0 = part * p.tiles_per_part, rt1 = min(p.row_tiles, rt0 + p.tiles_per_part);

    float ls[GAL_KMAX];                                       // thread q < 64: sorted list of query q (entries >= k stay sentinels)
    int li[GAL_KMAX];
#pragma unroll
    for (int i = 0; i < GAL_KMAX; ++i) { ls[i] = -INFINITY; li[i] = INT_MAX; }
    if (tid < BN) {
        float ts = -INFINITY; int ti = INT_MAX;                   // (-inf, INT_MAX): below every real entry, whatever the rows' norms (compareFaces does not clamp, face_recognizer.cpp:320-334)
        if (p.seed_i && n0 + tid < p.Q) {
            const size_t o = (size_t)(n0 + tid) * k + (k - 1);
            if (p.seed_i[o] >= 0) { ts = p.seed_s[o]; ti = p.seed_i[o]; }
        }
        cnt[tid] = 0; tau_s[tid] = ts; tau_i[tid] = ti;
    }
    if (tid == 0) overflow = 0;

    // query loader: pass i fills rows i*16 + (tid >> 4), slot tid & 15 <- source column (tid & 15) ^ (row & 15)
    const int qrow = tid >> 4;
    const float* const q_base = p.q + (size_t)(n0 + qrow) * K + (((tid & 15) ^ (qrow & 15)) * 4);
    const size_t q16 = (size_t)16 * K;
    const int fsw = fr & 15;

    for (int rt = rt0; rt < rt1; ++rt) {
        const long m0 = (long)rt * BM;
        const long myrow = m0 + wid * 32 + fr;
        const bool live = myrow < p.G;
        const float* a_ptr = (live ? p.gal + (size_t)myrow * K : p.zeros) + fh2 * 4;       // dead rows read the zero line (and are masked below)
        const int a_step = live ? 64 : 0;
        const float* q_src = q_base;
        v4f xa[2][8];
        auto load_a = [&](v4f (&x)[8]) {
#pragma unroll
            for (int s = 0; s < 8; ++s) x[s] = *reinterpret_cast<const v4f*>(a_ptr + s * 8);
            a_ptr += a_step;
        };
        v16f acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        auto multiply = [&](const v4f (&x)[8], int buf) {
            const v4f* Wt = ldsq[buf] + fr * 16;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int col = (2 * s + fh2) ^ fsw;
                v4f w[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) w[j] = Wt[j * 32 * 16 + col];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[s][e], w[j][e], acc[j], 0, 0, 0);
            }
        };
        // The query chunk goes global -> registers -> LDS (not by LDS-DMA): the compiler makes every LDS read wait for ALL vector-memory
        // traffic while an LDS-DMA is in flight (it cannot tell the two halves of ldsq apart: `s_waitcnt vmcnt(0)` in front of each
        // multiply, i.e. every chunk waited out its own gallery-row loads).  Through registers the dependencies are exact: the 4 query
        // loads are issued BEFORE the 8 row loads, so the ds_write after the multiply waits with vmcnt(8) and the row loads fly on until
        // the next barrier.  sched_barrier: left alone, the scheduler sinks the row loads below the multiply (one register set instead
        // of two) and serialises them with it.
        v4f qv[4];
        auto fetch_q = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const v4f*>(q_src + i * q16);
            q_src += 64;
        };
        auto store_q = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ldsq[buf][i * 256 + tid] = qv[i];
        };
        __syncthreads();                                   // previous tile's epilogue is done with the LDS lists / queues
        fetch_q();
        load_a(xa[0]);
        store_q(0);
        int kc = 0;
        for (; kc + 2 <= chunks; kc += 2) {
            __syncthreads();                               // queries of chunk kc are in LDS (and xa[0] has landed: the barrier drains vmcnt)
            fetch_q();
            load_a(xa[1]);
            __builtin_amdgcn_sched_barrier(0);
            multiply(xa[0], 0);
            store_q(1);
            __syncthreads();
            if (kc + 2 < chunks) { fetch_q(); load_a(xa[0]); }
            __builtin_amdgcn_sched_barrier(0);
            multiply(xa[1], 1);
            if (kc + 2 < chunks) store_q(0);
        }
        if (kc < chunks) {                                 // odd number of 64-deep chunks
            __syncthreads();
            multiply(xa[0], 0);
        }
        // ---- top-k epilogue (as gallery_topk_kernel)
        const long rbase = m0 + wid * 32 + 4 * fh2;
        auto push = [&](int g_lo, int g_hi) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int qi = j * 32 + fr;
                const float ts = tau_s[qi];
                const int ti = tau_i[qi];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if ((e >> 2) < g_lo || (e >> 2) >= g_hi) continue;
                    const long row = rbase + 8 * (e >> 2) + (e & 3);
                    const float sc = (acc[j][e] + 1.0f) / 2.0f;
                    const int gi = (int)(p.idx_base + row);
                    if (row < p.G && !gal_better(ts, ti, sc, gi)) {          // at least as good as the threshold entry (which may be this very row)
                        const int slot = atomicAdd(&cnt[qi], 1);
                        if (slot < GAL_QCAP) { que_s[qi][slot] = sc; que_i[qi][slot] = gi; }
                        else overflow = 1;
                    }
                }
            }
        };
        auto insert = [&]() {                                // thread q: its queue into its register list, then publish the new threshold
            if (tid < BN) {
                const int n = min(cnt[tid], GAL_QCAP);
                for (int c = 0; c < n; ++c) {
                    float s = que_s[tid][c];
                    int gi = que_i[tid][c];
#pragma unroll
                    for (int pos = 0; pos < GAL_KMAX; ++pos) {      // one compare-exchange pass keeps all 16 slots sorted (score desc, index asc)
                        const bool sw = gal_better(s, gi, ls[pos], li[pos]);
                        const float os = ls[pos]; const int oi = li[pos];
                        ls[pos] = sw ? s : os; li[pos] = sw ? gi : oi;
                        s = sw ? os : s; gi = sw ? oi : gi;
                    }
                }
                if (n > 0) {
                    int km1 = k - 1;
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(km1));                // (keeps the 16 position tests on the vector side: no SGPR mask per slot)
#endif
                    float ts = ls[0]; int ti = li[0];
#pragma unroll
                    for (int pos = 1; pos < GAL_KMAX; ++pos) { ts = pos == km1 ? ls[pos] : ts; ti = pos == km1 ? li[pos] : ti; }
                    // the local k-th entry bounds the global one as soon as the list holds k real rows; keep the tighter of it and the seed
                    if (ti != INT_MAX && gal_better(ts, ti, tau_s[tid], tau_i[tid])) { tau_s[tid] = ts; tau_i[tid] = ti; }
                }
                cnt[tid] = 0;
            }
        };
        push(0, 4);
        __syncthreads();
        if (overflow) {
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
    if (tid < BN && n0 + tid < p.Q) {
        const size_t o = ((size_t)part * p.Q + n0 + tid) * k;
#pragma unroll
        for (int pos = 0; pos < GAL_KMAX; ++pos)
            if (pos < k) { p.ps[o + pos] = li[pos] == INT_MAX ? -1.0f : ls[pos]; p.pi[o + pos] = li[pos] == INT_MAX ? -1 : li[pos]; }
    }
}

// parts the row range is cut into for a gallery of G rows and a query batch of Q (the caller sizes its partial-list buffers with it)
int gallery_parts(long G, int Q, int* tiles_per_part) {
    const int tiles_n = (Q + GAL_BN - 1) / GAL_BN;
    const long row_tiles = (G + GAL_BM - 1) / GAL_BM;
    const int slots = conv_num_cus() * 2;                       // 2 resident workgroups per CU (3 measured: no faster, and 768 lists per query leave the merge its slow path)
    long parts = slots / tiles_n;
    if (parts < 1) parts = 1;
    if (parts > row_tiles) parts = row_tiles;
    const long tpp = parts > 0 ? (row_tiles + parts - 1) / parts : 1;
    if (tiles_per_part) *tiles_per_part = (int)tpp;
    return (int)(tpp > 0 ? (row_tiles + tpp - 1) / tpp : 0);
}

// queries: packed [ceil64(Q)][dim] with zero rows behind Q; part_score / part_idx: [gallery_parts][Q][k]; seed_score / seed_idx: [Q][k] scratch.
// Two passes for a large gallery: the exact top-k of the first GAL_SEED_ROWS rows (same kernel + merge) gives every query an admission
// threshold, then the full scan runs with it — without the seed every workgroup spends its first tiles sorting rows that cannot matter.
void launch_gallery_topk(const float* gal, long G, int dim, const float* qpacked, int Q, int k, long idx_base, float* part_score, int* part_idx,
                         float* seed_score, int* seed_idx, hipStream_t s) {
    if (G <= 0 || Q <= 0) return;
    if (dim % 64 || k < 1 || k > GAL_KMAX) throw std::runtime_error("gallery: need dim % 64 == 0 and 1 <= k <= 16");
    if (idx_base + G > (long)INT_MAX) throw std::runtime_error("gallery: global row indices must fit in 31 bits");
    GalArgs a{};
    a.gal = gal; a.q = qpacked; a.zeros = conv_zero_line(); a.idx_base = idx_base; a.dim = dim; a.Q = Q; a.k = k;
    a.tiles_n = (Q + GAL_BN - 1) / GAL_BN;
    a.ps = part_score; a.pi = part_idx;
    constexpr long GAL_SEED_ROWS = 4096;
    static int seed_on = -1;
    if (seed_on < 0) { const char* e = getenv("FACEHIP_GAL_SEED"); seed_on = e ? atoi(e) : 1; }     // (0: no seed pass — A / B timing)
    if (seed_on && G >= 16 * GAL_SEED_ROWS && seed_score && seed_idx) {
        a.G = GAL_SEED_ROWS;
        a.row_tiles = (int)(GAL_SEED_ROWS / GAL_BM);
        const int sp = gallery_parts(a.G, Q, &a.tiles_per_part);
        hipLaunchKernelGGL(gallery_topk_kernel, dim3((unsigned)(sp * a.tiles_n)), dim3(256), 0, s, a);
        launch_topk_merge(part_score, part_idx, sp, Q, k, seed_score, seed_idx, s);
        a.seed_s = seed_score; a.seed_i = seed_idx;
    }
    a.G = G;
    a.row_tiles = (int)((G + GAL_BM - 1) / GAL_BM);
    const int parts = gallery_parts(G, Q, &a.tiles_per_part);
    hipLaunchKernelGGL(gallery_topk_kernel, dim3((unsigned)(parts * a.tiles_n)), dim3(256), 0, s, a);
}

}  // namespace fh
