// Micro-benchmark: how fast can a wave stream gallery rows straight into registers, as a function of how many different rows (= 128-byte
// lines per instruction) one wave-instruction touches?  hipcc --offload-arch=gfx950 -O3 row_stream.hip -o row_stream && ./row_stream
//   mode 0: 32 rows per wave, lane (row = l & 31, half = l >> 5) loads 16 B at k = (2 s + half) * 4   (gallery_scan_kernel's A fragment)
//   mode 1: 16 rows per wave, lane (row = l & 15, g = l >> 4) loads 16 B at k = (g + 4 s) * 4          (a 16x16x4 fragment)
//   mode 2:  8 rows per wave, lane (row = l & 7, g = l >> 3) loads 16 B at k = (g + 8 s) * 4           (128 B contiguous per row)
//   mode 3: fully coalesced: lane l loads 16 B at 16 l of a 1 KB block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512, 2) void stream(const float* __restrict__ g, long rows, int K, float* out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * 8 + wid, nw = (long)gridDim.x * 8;
    constexpr int RPW = MODE == 0 ? 32 : MODE == 1 ? 16 : MODE == 2 ? 8 : 1;   // rows per wave-step
    v4f acc = {0, 0, 0, 0};
    if (MODE == 3) {
        const long blocks = rows * K / 256;                                  // 1 KB blocks
        for (long b = wave * 8; b < blocks; b += nw * 8) {
            v4f x[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) x[s] = *reinterpret_cast<const v4f*>(g + (b + s) * 256 + lane * 4);
#pragma unroll
            for (int s = 0; s < 8; ++s) acc += x[s];
        }
    } else {
        const int r = lane % RPW, grp = lane / RPW, NG = 64 / RPW;            // NG 16-byte columns per instruction and row
        for (long r0 = wave * RPW; r0 < rows; r0 += nw * RPW) {
            const float* p = g + (r0 + r) * K + grp * 4;
            for (int c = 0; c < K; c += NG * 4 * 8) {                         // 8 loads in flight per step
                v4f x[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) x[s] = *reinterpret_cast<const v4f*>(p + c + s * NG * 4);
#pragma unroll
                for (int s = 0; s < 8; ++s) acc += x[s];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
int main() {
    const long rows = 1000000; const int K = 512;
    float* g; float* out;
    hipMalloc(&g, rows * K * 4 + 4096); hipMalloc(&out, 4);
    hipMemset(g, 0, rows * K * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        for (int grid : {256, 512}) {
            float best = 1e9;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(stream<0>, dim3(grid), dim3(512), 0, 0, g, rows, K, out);
                if (mode == 1) hipLaunchKernelGGL(stream<1>, dim3(grid), dim3(512), 0, 0, g, rows, K, out);
                if (mode == 2) hipLaunchKernelGGL(stream<2>, dim3(grid), dim3(512), 0, 0, g, rows, K, out);
                if (mode == 3) hipLaunchKernelGGL(stream<3>, dim3(grid), dim3(512), 0, 0, g, rows, K, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (it > 0 && ms < best) best = ms;
            }
            printf("mode %d grid %d (x 8 waves): %.3f ms = %.2f TB/s\n", mode, grid, best, rows * K * 4.0 / best / 1e9);
        }
    }
    return 0;
}
