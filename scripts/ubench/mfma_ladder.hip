// Ladder: which ingredient of the conv main loop costs MFMA throughput?
//  bit0: fragment ds_reads   bit1: barrier per chunk   bit2: 8 ds_write_b128 per chunk   bit3: 8 global loads per chunk
//  bit4: double LDS buffer alternate   bit5: ~96 dependent-free VALU ops per chunk   bit6: 8 LDS-DMA loads per chunk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(float* out, const float* in, const v4f* gsrc, int iters) {
    __shared__ v4f lds[2][2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[0][i] = v4f{in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]};
    __syncthreads();
    v16f acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    v4f a[2] = {lds[0][tid], lds[0][tid + 256]}, b[2] = {lds[0][tid + 512], lds[0][tid + 768]};
    v4f st[8];
    unsigned vjunk[8];
    for (int i = 0; i < 8; ++i) vjunk[i] = tid + i;
    for (int i = 0; i < 8; ++i) st[i] = lds[0][tid + i * 256];
    const v4f* g = gsrc + (size_t)blockIdx.x * 8192 + tid;
    for (int it = 0; it < iters; ++it) {
        const int buf = (MODE & 16) ? (it & 1) : 0;
        if (MODE & 32) {
#pragma unroll
            for (int i = 0; i < 96; ++i) vjunk[i & 7] = vjunk[i & 7] * 1664525u + (unsigned)it;
        }
        if (MODE & 64) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_global_load_lds((const float*)(g + ((it & 3) * 8 + i) * 256), (__attribute__((address_space(3))) void*)&lds[buf ^ 1][(tid >> 6) * 64 + i * 256], 16, 0, 0);
#endif
        }
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) st[i] = g[((it & 3) * 8 + i) * 256];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE & 1) {
                const int o = s * 256;
                a[0] = lds[buf][(o + tid) & 2047]; a[1] = lds[buf][(o + tid + 64) & 2047]; b[0] = lds[buf][(o + tid + 1024) & 2047]; b[1] = lds[buf][(o + tid + 1088) & 2047];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (MODE & 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) lds[buf ^ ((MODE & 16) ? 1 : 0)][tid + i * 256] = st[i];
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    for (int i = 0; i < 8; ++i) s += st[i][0] + (float)vjunk[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int OCC>
void run(const char* name, float* out, float* in, v4f* g, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, OCC>), dim3(grid), dim3(256), 0, 0, out, in, g, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    double fl = (double)grid * 4 * iters * 64 * 2 * 32 * 32 * 2;
    printf("%-60s grid %4d occ %d: %7.2f ms  %6.1f TF/s\n", name, grid, OCC, best, fl / best / 1e9);
}

int main() {
    float *out, *in; v4f* g;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 4096); hipMalloc(&g, (size_t)1024 * 8192 * 16 + 65536 * 16);
    hipMemset(g, 0, (size_t)1024 * 8192 * 16 + 65536 * 16);
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<0, 2>("mfma only", out, in, g, 512);
    run<1, 2>("+frag reads", out, in, g, 512);
    run<3, 2>("+frag reads +barrier", out, in, g, 512);
    run<7, 2>("+frag reads +barrier +ds_write x8", out, in, g, 512);
    run<23, 2>("+frag reads +barrier +ds_write x8 (2 buffers)", out, in, g, 512);
    run<15, 2>("+frag reads +barrier +ds_write x8 +global x8", out, in, g, 512);
    run<31, 2>("+frag reads +barrier +ds_write x8 +global x8 (2 buffers)", out, in, g, 512);
    run<9, 2>("+frag reads +global x8 (no barrier/no write)", out, in, g, 512);
    run<31, 2>("all, grid 256 (1 WG/CU)", out, in, g, 256);
    run<16 + 64 + 3, 2>("frag reads + barrier + LDS-DMA x8 (2 buffers)", out, in, g, 512);
    run<16 + 64 + 3 + 32, 2>("frag reads + barrier + LDS-DMA x8 + 96 VALU", out, in, g, 512);
    run<3 + 32, 2>("frag reads + barrier + 96 VALU", out, in, g, 512);
    run<1 + 32, 2>("frag reads + 96 VALU (no barrier)", out, in, g, 512);
    run<16 + 64 + 3 + 32, 2>("frag reads + barrier + LDS-DMA x8 + 96 VALU, 1 WG/CU", out, in, g, 256);
    return 0;
}
