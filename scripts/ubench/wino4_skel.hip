// Go / no-go skeleton for a FULLY fused Winograd F(4x4,3x3) layer at Cin = 128 (IResNet's 28x28 stage): what matrix-pipe share does the
// structure reach before any transform arithmetic exists?  One workgroup = 16 tiles x 64 output channels, 4 waves x (16 channels x 36
// frequencies x 16 tiles) = 144 accumulator registers per lane, v_mfma_f32_16x16x4_f32; per 16-deep k chunk and frequency a wave takes
// its U fragment (1 KB) from global memory (L2-resident weight image, fetched one frequency ahead) and the shared V fragment from LDS.
// (VALU / LW / GL template parameters: stand-ins for the transform's vector work, LDS stores and patch loads — the measured build
// (round 5) used the bare structure only: 512 workgroups (one round at two per CU) 54.8 us = 0.56 of the f32 MFMA peak, 784 workgroups
// (a 28x28x128->128 layer at B = 128) 97.4 us = 0.48, BEFORE any transform arithmetic: the 256 bytes of U per MFMA from L2 bind it.
// With the transforms on top the layer would not beat the 131 us of the unfused chain by enough to carry its complexity: not built.)
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/wino4_skel.hip -o scripts/ubench/wino4_skel && scripts/ubench/wino4_skel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int VALU, int LW, int GL>
__global__ __launch_bounds__(256, 2) void skel(const v4f* __restrict__ U, const v4f* __restrict__ D, float* __restrict__ out, int chunks) {
    extern __shared__ v4f Vdyn[];                          // [buffer][frequency][16 tiles x 4 k-quads]: 2 x 36.9 KB (dynamic: above the 64 KB static limit)
    v4f (*V)[36 * 64] = reinterpret_cast<v4f (*)[36 * 64]>(Vdyn);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2 * 36 * 64; i += 256) Vdyn[i] = D[(blockIdx.x * 97 + i) & 65535];
    __syncthreads();
    v4f acc[36];
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = v4f{0.f, 0.f, 0.f, 0.f};
    // weight image [chunk][f][wave][lane]: 36 KB per chunk and workgroup column tile; every workgroup streams the same 288 KB (L2)
    const v4f* u = U + wid * 64 + lane;
    v4f junk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) junk[i] = v4f{(float)tid, 1.f, 2.f, (float)i};
    const unsigned dbase = blockIdx.x * 4096u + tid;                      // (every index into D is masked to its 65 536 entries)
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        v4f a = u[0];
        v4f g[GL > 0 ? GL : 1];
#pragma unroll
        for (int i = 0; i < GL; ++i) g[i] = D[(dbase + (unsigned)(c * GL + i) * 256u) & 65535u];
#pragma unroll
        for (int f = 0; f < 36; ++f) {
            const v4f an = u[(f + 1 < 36 ? f + 1 : 0) * 256 + (f + 1 < 36 ? 0 : 36 * 256)];     // next frequency's fragment (next chunk's first at the end)
            const v4f b = V[buf][f * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[f], 0, 0, 0);
            // the stand-in transform work of this slot
#pragma unroll
            for (int i = 0; i < (VALU + 35) / 36; ++i) junk[(f + i) & 7] = junk[(f + i) & 7] * 1.0001f + junk[(f + i + 3) & 7];
            if (LW > 0 && f % (36 / (LW > 36 ? 36 : LW)) == 0) V[buf ^ 1][(f * 64 + tid) % (36 * 64)] = junk[f & 7] + (GL > 0 ? g[f % (GL > 0 ? GL : 1)] : junk[0]);
            a = an;
        }
        u += 36 * 256;
        __syncthreads();
    }
    v4f s = junk[0];
#pragma unroll
    for (int f = 0; f < 36; ++f) s += acc[f];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += junk[i];
    out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <int VALU, int LW, int GL>
void run(const v4f* U, const v4f* D, float* out, int blocks, const char* what) {
    const int chunks = 8;
    constexpr size_t LDS = 2 * 36 * 64 * sizeof(v4f);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&skel<VALU, LW, GL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS) != hipSuccess) { printf("attribute failed\n"); exit(1); }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((skel<VALU, LW, GL>), dim3(blocks), dim3(256), LDS, 0, U, D, out, chunks);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); exit(1); }
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((skel<VALU, LW, GL>), dim3(blocks), dim3(256), LDS, 0, U, D, out, chunks);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((skel<VALU, LW, GL>), dim3(blocks), dim3(256), LDS, 0, U, D, out, chunks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double flop = (double)blocks * 4 * chunks * 36 * 4 * 2.0 * 16 * 16 * 4;
    printf("%-44s blocks %4d  %7.1f us per launch  %6.1f TFLOP/s  (%.2f of 157.3)\n", what, blocks, ms * 1e3 / reps, flop * reps / (ms * 1e-3) / 1e12,
           flop * reps / (ms * 1e-3) / 157.3e12);
}

int main() {
    std::vector<float> h(4 * 65536 + 4 * (9 * 36 * 256 + 256));
    srand(2);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    v4f *U, *D; float* out;
    hipMalloc(&U, (9 * 36 * 256 + 256) * 16); hipMalloc(&D, 65536 * 16); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(U, h.data(), (9 * 36 * 256 + 256) * 16, hipMemcpyHostToDevice);
    hipMemcpy(D, h.data() + 4 * (9 * 36 * 256 + 256), 65536 * 16, hipMemcpyHostToDevice);
    for (int blocks : {512, 784}) {
        run<0, 0, 0>(U, D, out, blocks, "MFMA + weight stream + V fragment reads");
    }
    return 0;
}
