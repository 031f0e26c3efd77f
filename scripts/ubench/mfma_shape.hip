// Which f32 MFMA shape holds the higher clock in a GEMM-like loop?  v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32 at the same
// wave tile (32 rows x 128 columns, 64 accumulator registers), the same LDS fragment traffic (10 ds_read_b128 per 16 k) and the same
// FLOPs, on random data (MI355X_MICROARCH.md, DVFS give-back item 7: for bf16 the two shapes differ by 12-15 % in wall at equal cycles).
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_shape.hip -o scripts/ubench/mfma_shape && scripts/ubench/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int SHAPE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, const float* in, int iters) {
    __shared__ v4f lds[2][2048];                     // two "chunks": 128 A rows + 128 B rows x 32 k each (float4 units: row * 8 + k / 4)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 4096; i += 256) (&lds[0][0])[i] = v4f{in[(i * 4) & 65535], in[(i * 4 + 1) & 65535], in[(i * 4 + 2) & 65535], in[(i * 4 + 3) & 65535]};
    __syncthreads();
    float sum = 0.f;
    if (SHAPE == 32) {
        v16f acc[4];
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const int fr = lane & 31, fh2 = lane >> 5, fsw = (fr >> 1) & 7;
        for (int it = 0; it < iters; ++it) {
            const v4f* X = lds[it & 1] + (wid * 32 + fr) * 8;
            const v4f* W = lds[it & 1] + 1024 + fr * 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int col = (2 * s + fh2) ^ fsw;
                const v4f xv = X[col];
                v4f w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = W[j * 256 + col];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j][e], xv[e], acc[j], 0, 0, 0);
            }
        }
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) sum += acc[j][e];
    } else {
        v4f acc[2][8];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        const int r = lane & 15, kq = lane >> 4, sw = (r >> 1) & 7;
        for (int it = 0; it < iters; ++it) {
            const v4f* X = lds[it & 1] + (wid * 32 + r) * 8;
            const v4f* W = lds[it & 1] + 1024 + r * 8;
#pragma unroll
            for (int s = 0; s < 2; ++s) {              // 16 k per step
                const int col = (4 * s + kq) ^ sw;
                v4f xv[2], w[8];
#pragma unroll
                for (int i = 0; i < 2; ++i) xv[i] = X[i * 128 + col];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = W[j * 128 + col];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][e], xv[i][e], acc[i][j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) sum += acc[i][j][e];
    }
    out[blockIdx.x * 256 + tid] = sum;
}

template <int SHAPE, int OCC>
double run(float* out, const float* in, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<SHAPE, OCC>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<SHAPE, OCC>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double flop = 5.0 * blocks * 4.0 * iters * 2.0 * 32 * 128 * 32;       // per wave and iteration: 32 x 128 x 32 MACs
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    std::vector<float> h(65536);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out;
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {                  // interleaved rounds in one process (rule 24)
        printf("round %d  OCC2: 32x32x2 %.1f TF  16x16x4 %.1f TF   OCC1: 32x32x2 %.1f TF  16x16x4 %.1f TF\n", rep,
               run<32, 2>(out, in, 512, 4000), run<16, 2>(out, in, 512, 4000), run<32, 1>(out, in, 256, 4000), run<16, 1>(out, in, 256, 4000));
    }
    return 0;
}
