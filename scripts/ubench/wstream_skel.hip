// How much does the L2 -> register WEIGHT stream cost a 16x16x4-MFMA kernel whose B operand comes from LDS (the structure of conv_wino2.hip
// and of the fused F(4x4) skeleton wino4_skel.hip)?  Same loop, NF accumulators per tile block, TB tile blocks per wave: every U fragment
// (1 KB per wave, global -> registers one step ahead) feeds 4 TB MFMAs, i.e. 256 / TB bytes of weights per MFMA.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/wstream_skel.hip -o scripts/ubench/wstream_skel && scripts/ubench/wstream_skel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NF, int TB, int OCC, bool WLOAD, bool SHARE = false>
__global__ __launch_bounds__(256, OCC) void skel(const v4f* __restrict__ U, const v4f* __restrict__ D, float* __restrict__ out, int chunks) {
    extern __shared__ v4f Vdyn[];                          // [TB][NF][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < TB * NF * 64; i += 256) Vdyn[i] = D[(blockIdx.x * 97 + i) & 65535];
    __syncthreads();
    v4f acc[NF][TB];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < TB; ++t) acc[f][t] = v4f{0.f, 0.f, 0.f, 0.f};
    const v4f* u = U + (SHARE ? 0 : wid * 64) + lane;      // weight image [chunk][f][wave][lane]: NF KB per chunk and wave (SHARE: the four waves of a workgroup read the SAME fragments — L1 hits)
    v4f a = u[0];
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            v4f an = a;
            if (WLOAD) an = u[(f + 1) * 256];              // next fragment (the next chunk's first at f = NF - 1)
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const v4f b = Vdyn[(t * NF + f) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[f][t], 0, 0, 0);
            }
            a = an;
            __builtin_amdgcn_sched_barrier(0);             // (one fragment ahead, not all NF of them: keeps the register count honest)
        }
        u += NF * 256;
    }
    v4f s = a;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int t = 0; t < TB; ++t) s += acc[f][t];
    out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <int NF, int TB, int OCC, bool WLOAD, bool SHARE = false>
void run(const v4f* U, const v4f* D, float* out, const char* what) {
    const int chunks = 8 * 36 / NF;                        // the same number of fragments per wave in every variant
    const int blocks = 256 * OCC;                          // exactly one round
    const size_t lds = (size_t)TB * NF * 64 * sizeof(v4f);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&skel<NF, TB, OCC, WLOAD, SHARE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("attribute failed\n"); exit(1); }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((skel<NF, TB, OCC, WLOAD, SHARE>), dim3(blocks), dim3(256), lds, 0, U, D, out, chunks);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); exit(1); }
    const int reps = 20;
    (void)hipEventRecord(a);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((skel<NF, TB, OCC, WLOAD, SHARE>), dim3(blocks), dim3(256), lds, 0, U, D, out, chunks);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 4 * chunks * NF * TB * 4 * 2.0 * 16 * 16 * 4;
    printf("%-58s %7.1f us  %6.1f TFLOP/s (%.2f)  weights %3d B/MFMA\n", what, ms * 1e3 / reps, flop * reps / (ms * 1e-3) / 1e12,
           flop * reps / (ms * 1e-3) / 157.3e12, WLOAD ? 256 / TB : 0);
}

int main() {
    const size_t un = (size_t)(8 * 36 + 40) * 256;         // fragments: chunks * NF + slack for the look-ahead
    std::vector<float> h(4 * (un + 65536));
    srand(3);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    v4f *U, *D; float* out;
    (void)hipMalloc(&U, un * 16); (void)hipMalloc(&D, 65536 * 16); (void)hipMalloc(&out, 1024 * 256 * 4);
    (void)hipMemcpy(U, h.data(), un * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(D, h.data() + 4 * un, 65536 * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<36, 1, 2, true>(U, D, out, "36 acc, 1 tile block, 2 workgroups per CU");
        run<36, 1, 2, false>(U, D, out, "   ... without the weight loads");
        run<36, 1, 1, true>(U, D, out, "36 acc, 1 tile block, 1 workgroup per CU");
        run<36, 2, 1, true>(U, D, out, "36 acc, 2 tile blocks, 1 workgroup per CU");
        run<36, 2, 1, false>(U, D, out, "   ... without the weight loads");
        run<16, 1, 2, true>(U, D, out, "16 acc (F(2x2)), 1 tile block, 2 workgroups per CU");
        run<16, 1, 2, true, true>(U, D, out, "   ... the 4 waves of a workgroup share their fragments");
        run<16, 2, 2, true>(U, D, out, "16 acc, 2 tile blocks, 2 workgroups per CU");
        run<16, 4, 1, true>(U, D, out, "16 acc, 4 tile blocks, 1 workgroup per CU");
        run<16, 4, 1, false>(U, D, out, "   ... without the weight loads");
    }
    return 0;
}
