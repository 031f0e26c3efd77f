// f32 MFMA peak calibration: registers-only, and with our LDS read pattern (1 ds_read_b128 per 4 MFMA per operand pair).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters, unsigned long long* clk) {
    __shared__ v4f lds[4096];
    const int tid = threadIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();     // shader clock vs the constant 100 MHz counter
    for (int i = tid; i < 4096; i += 256) lds[i] = v4f{in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]};
    __syncthreads();
    v16f acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    v4f a[2] = {lds[tid], lds[tid + 256]}, b[2] = {lds[tid + 512], lds[tid + 768]};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const int o = (it & 3) * 1024;
            a[0] = lds[o + tid]; a[1] = lds[o + tid + 256]; b[0] = lds[o + tid + 512]; b[1] = lds[o + tid + 768];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0 && clk) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 4096);
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk; hipMalloc(&clk, 16);
    for (int iters : {20000, 400000, 4000000})            // ~1.3 ms, ~27 ms, ~270 ms: does the rate hold once power management reacts?
    for (int mode = 0; mode < 2; ++mode)
        for (int grid : {512}) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, in, iters, clk);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, in, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
                double fl = (double)grid * 4 * iters * 16 * 2 * 32 * 32 * 2;
                if (rep == 2) printf("iters %d mode %d (%s) grid %d: %.2f ms  %.1f TF/s  shader clock %.0f MHz\n", iters, mode, mode ? "ds_read_b128 x4 per 16 mfma" : "registers only", grid, ms, fl / ms / 1e9, (double)c[0] / ((double)c[1] / 100.0));
            }
        }
    return 0;
}
