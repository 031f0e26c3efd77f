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
#include <hip/hip_runtime.h>

#include <algorithm>
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
    int dbg;                // tuning only (FACEHIP_GAL_DBG): 1 = no top-k epilogue, 2 = no MFMAs, 4 = no row loads
    unsigned* tau_g;        // gallery_scan_kernel: [tiles_n * 64] admission thresholds shared by ALL workgroups (order-preserving uint keys)
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 3: the scan with the QUERIES RESIDENT IN LDS and NO barrier between the prologue and the final list merge.
//
// gallery_topk_kernel above re-loads the 64-query tile for every 128 gallery rows (16 KB per 64-deep chunk through registers and
// ds_write, two barriers per chunk), runs its top-k update behind two more barriers per tile, and needs a seed pass (a launch of its
// own over the first 4 096 rows + a merge) to start with closed lists.  Measured on this kernel's first form (queries resident, but the
// old epilogue): the seed launch 196 us and the per-tile barriers ~240 us of a 0.86 ms call — with one 8-wave workgroup per CU a
// barrier stalls the whole CU on its slowest wave's memory latency.  Now:
//   * queries: at dim <= 512 the whole tile fits LDS once (64 x 512 x 4 B = 128 KB of the CU's 160): [64 rows][dim / 4 float4], 16-byte
//     column XOR-swizzled by (row & 15) so that the B-fragment reads of 16 neighbouring rows spread over all banks; read as fragments
//     one 8-deep step AHEAD of their MFMAs (two register sets, sched_barrier).
//   * gallery rows: each wave streams ITS OWN 32 rows straight into registers, one 64-deep chunk (8 loads, 32 VGPRs) ahead of the one
//     being multiplied, the next tile's first chunk issued before the epilogue.
//   * top-k WITHOUT synchronisation: every wave keeps its own sorted list per query, lane l = query l (64 lanes, 64 queries).  The 32x32
//     accumulator of query block j gives lane (fr, h) the scores of query 32 j + fr on ITS 16 rows; lane l's own query is 32 h + fr, so
//     per accumulator position one lane^32 exchange (ds_bpermute, no LDS memory) hands every lane the partner's score for its query.
//     A candidate is inserted by one compare-exchange pass over the 16 register slots.  Admission threshold: one float per query in
//     LDS, raised by any wave to the k-th score of its own full list (a plain store of a larger value: racy, but every value ever
//     stored is a valid bound — k rows at least that good exist) and read unsynchronised; scores >= it are candidates, the exact
//     (score desc, index asc) order is decided at insertion.  No seed pass: a wave's list fills on its first tile and the insertions
//     die out like k ln(n / k).
//   * at the end the eight lists per query meet in LDS (the query image is no longer needed) and threads 0..63 merge them: the
//     kernel's output is one list per workgroup and query, as before.
constexpr int GS_WAVES = 8, GS_BM = GS_WAVES * 32;
// float <-> unsigned key with the same order (so that an integer atomicMax is a float max): negative floats flip all bits, others the sign
__device__ __forceinline__ unsigned gal_key(float f) { const unsigned u = __builtin_bit_cast(unsigned, f); return (u & 0x80000000u) ? ~u : u | 0x80000000u; }
__device__ __forceinline__ float gal_unkey(unsigned k) { return __builtin_bit_cast(float, (k & 0x80000000u) ? k & 0x7fffffffu : ~k); }

__global__ __launch_bounds__(GS_WAVES * 64, 2) void gallery_scan_kernel(const GalArgs p) {
    constexpr int BN = GAL_BN, TN = BN / 32;
    extern __shared__ v4f gsm[];
    const int K = p.dim, K4 = K >> 2, chunks = K / 64, k = p.k;
    v4f* const Ql = gsm;                                                   // [64][K4], column ^ (row & 15); later: the waves' lists
    float* const tau = reinterpret_cast<float*>(Ql + BN * K4);             // [64] admission threshold (score of SOME wave's k-th entry)
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
    const int rt0 = part * p.tiles_per_part, rt1 = min(p.row_tiles, rt0 + p.tiles_per_part);

    float ls[GAL_KMAX];                                                    // this wave's sorted list of query `lane` (entries >= k stay sentinels)
    int li[GAL_KMAX];
#pragma unroll
    for (int i = 0; i < GAL_KMAX; ++i) { ls[i] = -INFINITY; li[i] = INT_MAX; }
    {   // queries -> LDS: all loads of a thread first, then the writes (a load -> write chain per float4 costs a round trip each)
        constexpr int QPT = 16;                                            // float4 per thread and pass: 512 threads x 16 = one 64 x 512 tile
        for (int base = 0; base < BN * K4; base += GS_WAVES * 64 * QPT) {
            v4f qv[QPT];
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const int i = min(base + u * GS_WAVES * 64 + tid, BN * K4 - 1);
                const int q = i / K4, c = i - q * K4;
                qv[u] = *reinterpret_cast<const v4f*>(p.q + (size_t)(n0 + q) * K + 4 * c);
            }
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const int i = base + u * GS_WAVES * 64 + tid;
                if (i < BN * K4) { const int q = i / K4, c = i - q * K4; Ql[q * K4 + (c ^ (q & 15))] = qv[u]; }
            }
        }
    }
    if (tid < BN) tau[tid] = -INFINITY;
    __syncthreads();

    const v4f* const qrow = Ql + fr * K4;                                  // + j * 32 * K4 + (col ^ swz)
    const int swz = fr & 15;                                               // (rows fr and fr + 32 share it)
    v4f xa[2][8];
    const float* a_ptr = p.zeros;
    int a_step = 0;
    auto row_setup = [&](int rt) __attribute__((always_inline)) {
        const long myrow = (long)rt * GS_BM + wid * 32 + fr;
        const bool live = myrow < p.G;
        a_ptr = (live && !(p.dbg & 4) ? p.gal + (size_t)myrow * K : p.zeros) + fh2 * 4;     // dead rows read the zero line (and are masked in the epilogue)
        a_step = live && !(p.dbg & 4) ? 64 : 0;
    };
    auto load_a = [&](v4f (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) x[s] = *reinterpret_cast<const v4f*>(a_ptr + s * 8);
        a_ptr += a_step;
    };
    if (rt0 < rt1) { row_setup(rt0); load_a(xa[0]); }
    for (int rt = rt0; rt < rt1; ++rt) {
        const long m0 = (long)rt * GS_BM;
        // the chip-wide threshold of this lane's query (any workgroup's published k-th score: after the first round of tiles it is the
        // top-k of >= 60 000 rows, and hardly anything passes any more).  Loaded now, used in the epilogue: its latency hides in the K loop.
        const unsigned tg_key = __hip_atomic_load(p.tau_g + n0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v16f acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        v4f wq[2][TN];
        auto qfrag = [&](int gs) __attribute__((always_inline)) {          // global step index gs = 8 * chunk + s
            const int col = (2 * gs + fh2) ^ swz;
#pragma unroll
            for (int j = 0; j < TN; ++j) wq[gs & 1][j] = qrow[j * 32 * K4 + col];
        };
        auto multiply = [&](const v4f (&x)[8], int kc) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int gs = kc * 8 + s;
                if (gs + 1 < chunks * 8) qfrag(gs + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (p.dbg & 2) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j][s] += x[s][0] * wq[gs & 1][j][0] + x[s][1] * wq[gs & 1][j][1] + x[s][2] + x[s][3];
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[s][e], wq[gs & 1][j][e], acc[j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        qfrag(0);
        int kc = 0;
        for (; kc + 2 <= chunks; kc += 2) {
            load_a(xa[1]);
            __builtin_amdgcn_sched_barrier(0);                              // the row loads stay ABOVE the multiply (two register sets in flight)
            multiply(xa[0], kc);
            if (kc + 2 < chunks) load_a(xa[0]);
            else if (rt + 1 < rt1) { row_setup(rt + 1); load_a(xa[0]); }    // the next tile's first chunk flies through the epilogue
            __builtin_amdgcn_sched_barrier(0);
            multiply(xa[1], kc + 1);
        }
        if (kc < chunks) {                                                  // odd number of 64-deep chunks
            multiply(xa[0], kc);
            if (rt + 1 < rt1) { row_setup(rt + 1); load_a(xa[0]); }
        }
        // ---- top-k epilogue, wave-private.  C/D map: acc[j][e] = query 32 j + fr, row rbase(h) + 8 (e >> 2) + (e & 3), rbase(h) = .. + 4 h
        const long rb0 = m0 + wid * 32;
        {
            const float tg = gal_unkey(tg_key);
            if (tg > tau[lane]) tau[lane] = tg;                             // (same wave reads it back below: LDS operations of a wave stay in order)
        }
        const float th0 = tau[fr], th1 = tau[32 + fr];
        unsigned pass = 0;                                                  // bit e: acc[0][e] is a candidate, bit 16 + e: acc[1][e]
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool dead = rb0 + 4 * fh2 + 8 * (e >> 2) + (e & 3) >= p.G;
            acc[0][e] = dead ? -INFINITY : (acc[0][e] + 1.0f) / 2.0f;       // compareFaces' mapped score, as the reference computes it
            acc[1][e] = dead ? -INFINITY : (acc[1][e] + 1.0f) / 2.0f;
            pass |= acc[0][e] >= th0 ? 1u << e : 0u;
            pass |= acc[1][e] >= th1 ? 1u << (16 + e) : 0u;
        }
        pass &= p.dbg & 1 ? 0u : ~0u;
        if (__builtin_amdgcn_ballot_w64(pass != 0) != 0) {                  // rare once the lists have closed
            const float mine_th = fh2 ? th1 : th0;
            auto insert = [&](float s_, int gi) __attribute__((always_inline)) {
#pragma unroll
                for (int pos = 0; pos < GAL_KMAX; ++pos) {                  // one compare-exchange pass keeps all 16 slots sorted (score desc, index asc)
                    const bool sw = gal_better(s_, gi, ls[pos], li[pos]);
                    const float os = ls[pos]; const int oi = li[pos];
                    ls[pos] = sw ? s_ : os; li[pos] = sw ? gi : oi;
                    s_ = sw ? os : s_; gi = sw ? oi : gi;
                }
            };
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned bits = (1u << e) | (1u << (16 + e));
                if (__builtin_amdgcn_ballot_w64((pass & bits) != 0) == 0) continue;
                // my query is 32 h + fr: my own score for it is acc[h][e] (my row); the partner lane (^32) holds it for ITS row in acc[h][e] too,
                // i.e. I send the score I hold for the partner's query, acc[1 - h][e]
                const float own = fh2 ? acc[1][e] : acc[0][e];
                const float snd = fh2 ? acc[0][e] : acc[1][e];
                const float got = __shfl_xor(snd, 32);
                const int r_own = (int)(p.idx_base + rb0 + 4 * fh2 + 8 * (e >> 2) + (e & 3));
                const int r_got = (int)(p.idx_base + rb0 + 4 * (1 - fh2) + 8 * (e >> 2) + (e & 3));
                const bool c_own = own >= mine_th, c_got = got >= mine_th;
                if (__builtin_amdgcn_ballot_w64(c_own) != 0) { if (c_own) insert(own, r_own); }
                if (__builtin_amdgcn_ballot_w64(c_got) != 0) { if (c_got) insert(got, r_got); }
            }
            // publish: the k-th entry of a full list bounds the global k-th from below
            int km1 = k - 1;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(km1));                                   // (keeps the 16 position tests on the vector side)
#endif
            float ts = ls[0]; int ti = li[0];
#pragma unroll
            for (int pos = 1; pos < GAL_KMAX; ++pos) { ts = pos == km1 ? ls[pos] : ts; ti = pos == km1 ? li[pos] : ti; }
            if (ti != INT_MAX && ts > tau[lane]) {
                tau[lane] = ts;
                atomicMax(p.tau_g + n0 + lane, gal_key(ts));
            }
        }
    }
    // ---- the eight waves' lists of a query -> one list per workgroup
    __syncthreads();                                                        // every wave is done with the query image
    float* const Ls = reinterpret_cast<float*>(Ql);                         // [8 waves][64 queries][16]
    int* const Li = reinterpret_cast<int*>(Ls + GS_WAVES * BN * GAL_KMAX);
#pragma unroll
    for (int pos = 0; pos < GAL_KMAX; ++pos) { Ls[(wid * BN + lane) * GAL_KMAX + pos] = ls[pos]; Li[(wid * BN + lane) * GAL_KMAX + pos] = li[pos]; }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Q) {
        int head[GS_WAVES];
#pragma unroll
        for (int w = 0; w < GS_WAVES; ++w) head[w] = 0;
        const size_t o = ((size_t)part * p.Q + n0 + tid) * k;
        for (int pos = 0; pos < k; ++pos) {                                 // k rounds of "best head of the eight sorted lists"
            float bs = -INFINITY; int bi = INT_MAX, bw = -1;
#pragma unroll
            for (int w = 0; w < GS_WAVES; ++w) {
                const int hd = min(head[w], GAL_KMAX - 1);
                const float s_ = Ls[(w * BN + tid) * GAL_KMAX + hd];
                const int i_ = Li[(w * BN + tid) * GAL_KMAX + hd];
                if (head[w] < GAL_KMAX && i_ != INT_MAX && (bw < 0 || gal_better(s_, i_, bs, bi))) { bs = s_; bi = i_; bw = w; }
            }
#pragma unroll
            for (int w = 0; w < GS_WAVES; ++w) head[w] += w == bw ? 1 : 0;
            p.ps[o + pos] = bw < 0 ? -1.0f : bs;
            p.pi[o + pos] = bw < 0 ? -1 : bi;
        }
    }
}

static bool gallery_scan_ok(int dim) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("FACEHIP_GALLERY_SCAN"); on = e ? atoi(e) : 1; }   // (0 = the round-2 kernel: A / B timing)
    return on && dim <= 512;
}
static size_t gallery_scan_lds(int dim) {                                  // the query image (>= the 64 KB the final lists need) + the thresholds
    return std::max((size_t)GAL_BN * (dim / 4) * 16, (size_t)GS_WAVES * GAL_BN * GAL_KMAX * 8) + (size_t)GAL_BN * 4;
}

// parts the row range is cut into for a gallery of G rows and a query batch of Q (the caller sizes its partial-list buffers with it)
int gallery_parts(long G, int Q, int dim, int* tiles_per_part) {
    const bool scan = gallery_scan_ok(dim);
    const int tiles_n = (Q + GAL_BN - 1) / GAL_BN;
    const long bm = scan ? GS_BM : GAL_BM;
    const long row_tiles = (G + bm - 1) / bm;
    // resident workgroups: one 8-wave workgroup per CU (queries resident in LDS), or 2 per CU for the round-2 kernel (3 measured: no
    // faster, and 768 lists per query leave the merge its slow path)
    const int slots = conv_num_cus() * (scan ? 1 : 2);
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
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("FACEHIP_GAL_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
    constexpr long GAL_SEED_ROWS = 4096;
    const bool scan = gallery_scan_ok(dim);
    const long bm = scan ? GS_BM : GAL_BM;
    const size_t lds = scan ? gallery_scan_lds(dim) : 0;
    if (scan) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gallery_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
    }
    auto launch = [&](int parts) {
        if (scan) {
            // chip-wide thresholds start at key 0 (below every float): the caller's seed_score buffer holds >= ceil64(Q) words
            a.tau_g = reinterpret_cast<unsigned*>(seed_score);
            (void)hipMemsetAsync(a.tau_g, 0, (size_t)a.tiles_n * GAL_BN * sizeof(unsigned), s);
            hipLaunchKernelGGL(gallery_scan_kernel, dim3((unsigned)(parts * a.tiles_n)), dim3(GS_WAVES * 64), lds, s, a);
        }
        else hipLaunchKernelGGL(gallery_topk_kernel, dim3((unsigned)(parts * a.tiles_n)), dim3(256), 0, s, a);
    };
    if (scan && !seed_score) throw std::runtime_error("gallery: the scan kernel needs the seed_score scratch (>= ceil64(Q) words)");
    if (!scan && G >= 16 * GAL_SEED_ROWS && seed_score && seed_idx) {   // (gallery_scan_kernel needs no seed pass)
        a.G = GAL_SEED_ROWS;
        a.row_tiles = (int)(GAL_SEED_ROWS / bm);
        const int sp = gallery_parts(a.G, Q, dim, &a.tiles_per_part);
        launch(sp);
        launch_topk_merge(part_score, part_idx, sp, Q, k, seed_score, seed_idx, s);
        a.seed_s = seed_score; a.seed_i = seed_idx;
    }
    a.G = G;
    a.row_tiles = (int)((G + bm - 1) / bm);
    const int parts = gallery_parts(G, Q, dim, &a.tiles_per_part);
    launch(parts);
}

}  // namespace fh
