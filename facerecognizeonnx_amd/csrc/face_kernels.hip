// face_kernels.hip — the non-network stages of the detect -> align -> embed -> compare path as
// HBM-bound / integer kernels for gfx950.  Each kernel names the reference lines it replaces.
// Compiled with -ffp-contract=off: the fp64 similarity estimate and the fixed-point image
// arithmetic are meant to be bit-identical to the CPU oracle (oracle/face_oracle.c).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace fh {

typedef float v4f __attribute__((ext_vector_type(4)));

static inline int grid_for(long n, int block = 256, int cap = 256 * 16) {
    long b = (n + block - 1) / block;
    return (int)(b < 1 ? 1 : b > cap ? cap : b);
}

// ------------------------------------------------------------------------------------------
// FaceDetector::preprocess / FaceRecognizer::preprocess
//   (src/face_detector.cpp:120-136, src/face_recognizer.cpp:135-150)
// zero letterbox (top-left paste), BGR->RGB, (v - 127.5f) / 128.0f, written channels-last with a
// zero 4th lane so the first convolution reads one float4 per pixel.  Exact in fp32.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ src, long img_stride, int srcH, int srcW,
                                                         int step, int B, int outH, int outW, float* __restrict__ out) {
    const long total = (long)B * outH * outW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % outW);
        const long r = i / outW;
        const int y = (int)(r % outH);
        const int b = (int)(r / outH);
        float bb = 0.f, gg = 0.f, rr = 0.f;
        if (y < srcH && x < srcW) {
            const uint8_t* p = src + (size_t)b * img_stride + (size_t)y * step + (size_t)x * 3;
            bb = (float)p[0]; gg = (float)p[1]; rr = (float)p[2];
        }
        v4f o;
        o[0] = (rr - 127.5f) / 128.0f;
        o[1] = (gg - 127.5f) / 128.0f;
        o[2] = (bb - 127.5f) / 128.0f;
        o[3] = 0.f;
        *reinterpret_cast<v4f*>(out + i * 4) = o;
    }
}

void launch_det_preprocess(const uint8_t* frames, long img_stride, int rows, int cols, int step, int B, int inH, int inW,
                           int newH, int newW, float* out, hipStream_t s) {
    (void)rows; (void)cols;
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long)B * inH * inW)), dim3(256), 0, s, frames, img_stride, newH, newW,
                       step, B, inH, inW, out);
}

void launch_rec_preprocess(const uint8_t* crops, int n, int H, int W, float* out, hipStream_t s) {
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long)n * H * W)), dim3(256), 0, s, crops, (long)H * W * 3, H, W, W * 3, n,
                       H, W, out);
}

// ------------------------------------------------------------------------------------------
// OpenCV helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_round_d(double v) { return __double2int_rn(v); }     // half-to-even, saturating
__device__ __forceinline__ int cv_floor_f(float v) { return (int)floorf(v); }
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
__device__ __forceinline__ int sat_short_f(float v) { return sat_short(__float2int_rn(v)); }

// One output sample of cv::resize(INTER_LINEAR, CV_8UC3) — classic 11-bit fixed-point path,
// same-size copy and the 2x -> INTER_AREA dispatch included (oracle: orc_resize_bilinear_u8c3).
__device__ int resize_px(const uint8_t* __restrict__ src, int sh, int sw, int sstep, int dh, int dw, int dx, int dy, int c) {
    if (dh == sh && dw == sw) return src[(size_t)dy * sstep + dx * 3 + c];
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
    const int isx = cv_round_d(scale_x), isy = cv_round_d(scale_y);
    const bool area_fast = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    if (area_fast && isx == 2 && isy == 2) {
        const uint8_t* s0 = src + (size_t)(2 * dy) * sstep;
        const uint8_t* s1 = s0 + sstep;
        return (s0[6 * dx + c] + s0[6 * dx + 3 + c] + s1[6 * dx + c] + s1[6 * dx + 3 + c] + 2) >> 2;
    }
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    bool edge = false;
    if (sx + 1 >= sw) { edge = true; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
    const int a0 = sat_short_f((1.f - fx) * 2048.f), a1 = sat_short_f(fx * 2048.f);
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    const int b0 = sat_short_f((1.f - fy) * 2048.f), b1 = sat_short_f(fy * 2048.f);
    int y0 = sy, y1 = sy + 1;
    y0 = y0 >= 0 ? (y0 < sh ? y0 : sh - 1) : 0;
    y1 = y1 >= 0 ? (y1 < sh ? y1 : sh - 1) : 0;
    const uint8_t* r0 = src + (size_t)y0 * sstep + sx * 3 + c;
    const uint8_t* r1 = src + (size_t)y1 * sstep + sx * 3 + c;
    const int S0 = edge ? r0[0] * 2048 : r0[0] * a0 + r0[3] * a1;
    const int S1 = edge ? r1[0] * 2048 : r1[0] * a0 + r1[3] * a1;
    const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    return v < 0 ? 0 : v > 255 ? 255 : v;
}

__global__ __launch_bounds__(256) void resize_kernel(const uint8_t* __restrict__ src, long src_stride, int sh, int sw, int sstep,
                                                     uint8_t* __restrict__ dst, long dst_stride, int dh, int dw, int dstep, int n) {
    const long total = (long)n * dh * dw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dx = (int)(i % dw);
        const long r = i / dw;
        const int dy = (int)(r % dh);
        const int b = (int)(r / dh);
        const uint8_t* s = src + (size_t)b * src_stride;
        uint8_t* d = dst + (size_t)b * dst_stride + (size_t)dy * dstep + dx * 3;
        for (int c = 0; c < 3; ++c) d[c] = (uint8_t)resize_px(s, sh, sw, sstep, dh, dw, dx, dy, c);
    }
}

void launch_resize_u8c3(const uint8_t* src, long src_stride, int sh, int sw, int sstep, uint8_t* dst, long dst_stride, int dh,
                        int dw, int dstep, int n, hipStream_t s) {
    hipLaunchKernelGGL(resize_kernel, dim3(grid_for((long)n * dh * dw)), dim3(256), 0, s, src, src_stride, sh, sw, sstep, dst,
                       dst_stride, dh, dw, dstep, n);
}

// ------------------------------------------------------------------------------------------
// Anchor decode (the step the reference lacks) + FaceDetector::postprocess row
// loop (src/face_detector.cpp:249-278): strict score > thr, /scale, int truncation, width from
// the float difference.  Payload goes to cand[frame][anchor]; the surviving anchors' sort keys
// (score descending, anchor index ascending — the total order this build fixes for the
// reference's unstable std::sort, :357) are compacted with one atomic per survivor.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx) {
    unsigned u = __float_as_uint(score);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // ascending-orderable
    return ((unsigned long long)(~u) << 32) | idx;          // descending score, ascending index
}

__device__ __forceinline__ void emit_face(const float* o15, float scale, FaceRec* f) {
    const float x1 = o15[0] / scale, y1 = o15[1] / scale, x2 = o15[2] / scale, y2 = o15[3] / scale;
    f->x = (int)x1; f->y = (int)y1; f->w = (int)(x2 - x1); f->h = (int)(y2 - y1);
    f->score = o15[4];
#pragma unroll
    for (int j = 0; j < 10; ++j) f->lm[j] = o15[5 + j] / scale;
}

__global__ __launch_bounds__(256) void scrfd_decode_kernel(const DecodeArgs a) {
    const int gw8 = a.inW / 8, gh8 = a.inH / 8, gw16 = a.inW / 16, gh16 = a.inH / 16, gw32 = a.inW / 32, gh32 = a.inH / 32;
    const int n8 = gw8 * gh8 * 2, n16 = gw16 * gh16 * 2, n32 = gw32 * gh32 * 2;
    const int N = n8 + n16 + n32;
    const int b = blockIdx.y;                              // one grid row per frame: no 64-bit index arithmetic per anchor
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < N; r += gridDim.x * blockDim.x) {
        int si, i, gw, s, ns;
        if (r < n8) { si = 0; i = r; gw = gw8; s = 8; ns = n8; }
        else if (r < n8 + n16) { si = 1; i = r - n8; gw = gw16; s = 16; ns = n16; }
        else { si = 2; i = r - n8 - n16; gw = gw32; s = 32; ns = n32; }
        const float score = a.score[si][(size_t)b * ns + i];
        if (!(score > a.thr)) continue;
        const int cell = i >> 1;
        const int gy = cell / gw, gx = cell - gy * gw;
        const float cx = (float)(gx * s), cy = (float)(gy * s), fs = (float)s;
        const float* d = a.bbox[si] + ((size_t)b * ns + i) * 4;
        const float* k = a.kps[si] + ((size_t)b * ns + i) * 10;
        float o[15];
        o[0] = cx - d[0] * fs; o[1] = cy - d[1] * fs; o[2] = cx + d[2] * fs; o[3] = cy + d[3] * fs;
        o[4] = score;
#pragma unroll
        for (int j = 0; j < 5; ++j) { o[5 + 2 * j] = cx + k[2 * j] * fs; o[6 + 2 * j] = cy + k[2 * j + 1] * fs; }
        emit_face(o, a.scale, a.cand + (size_t)b * a.cap + r);
        const int pos = atomicAdd(a.count + b, 1);
        if (pos < a.cap) a.keys[(size_t)b * a.cap + pos] = make_key(score, (unsigned)r);
    }
}

void launch_scrfd_decode(const DecodeArgs& a, hipStream_t s) {
    const int N = ((a.inW / 8) * (a.inH / 8) + (a.inW / 16) * (a.inH / 16) + (a.inW / 32) * (a.inH / 32)) * 2;
    if (N <= 0 || a.B <= 0) return;
    hipLaunchKernelGGL(scrfd_decode_kernel, dim3((N + 255) / 256, a.B), dim3(256), 0, s, a);
}

// The reference's own layout: rows [n, feat >= 15] = x1,y1,x2,y2,score,kps (src/face_detector.cpp:242-325)
__global__ __launch_bounds__(256) void rows_threshold_kernel(const float* __restrict__ rows, int B, int n, int feat, float scale,
                                                             float thr, FaceRec* cand, unsigned long long* keys, int* count, int cap) {
    const long total = (long)B * n;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int b = (int)(t / n);
        const int r = (int)(t - (long)b * n);
        const float* o = rows + (size_t)t * feat;
        const float score = o[4];
        if (!(score > thr) || r >= cap) continue;
        float o15[15];
#pragma unroll
        for (int j = 0; j < 15; ++j) o15[j] = o[j];
        emit_face(o15, scale, cand + (size_t)b * cap + r);
        const int pos = atomicAdd(count + b, 1);
        if (pos < cap) keys[(size_t)b * cap + pos] = make_key(score, (unsigned)r);
    }
}

void launch_rows_threshold(const float* rows, int B, int n, int feat, float scale, float thr, FaceRec* cand,
                           unsigned long long* keys, int* count, int cap, hipStream_t s) {
    hipLaunchKernelGGL(rows_threshold_kernel, dim3(grid_for((long)B * n)), dim3(256), 0, s, rows, B, n, feat, scale, thr, cand, keys,
                       count, cap);
}

// ------------------------------------------------------------------------------------------
// FaceDetector::iou + FaceDetector::nms (src/face_detector.cpp:340-384): integer intersection,
// integer denominator, one float divide, strict  iou > thr;  greedy in score order.
// One workgroup per frame: bitonic sort of the 64-bit keys (LDS for <= 2048 candidates, in
// place in global memory otherwise), then the greedy sweep with all lanes testing one pivot
// against the remaining boxes, then an ordered compaction of the survivors.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float iou_int(int4 a, int4 b) {
    const int x1 = max(a.x, b.x), y1 = max(a.y, b.y);
    const int x2 = min(a.x + a.z, b.x + b.z), y2 = min(a.y + a.w, b.y + b.w);
    const int w = max(0, x2 - x1), h = max(0, y2 - y1);
    const int inter = w * h;
    const int area1 = a.z * a.w, area2 = b.z * b.w;
    return (float)inter / (float)(area1 + area2 - inter);
}

constexpr int NMS_T = 1024;
constexpr int NMS_SMALL = 2048;

__global__ __launch_bounds__(NMS_T) void sort_nms_kernel(const FaceRec* __restrict__ cand, unsigned long long* __restrict__ keys_g,
                                                         const int* __restrict__ count, int cap, float thr, FaceRec* __restrict__ out,
                                                         int* __restrict__ out_count, int max_out, int* __restrict__ ws) {
    __shared__ unsigned long long skeys[NMS_SMALL];
    __shared__ int4 sbox[NMS_SMALL];
    __shared__ unsigned char ssup[NMS_SMALL];
    __shared__ int wave_tot[NMS_T / 64];
    __shared__ int run_base;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(count[b], cap);
    const FaceRec* fc = cand + (size_t)b * cap;
    if (n <= 0) { if (tid == 0) out_count[b] = 0; return; }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    const bool small = np2 <= NMS_SMALL;
    unsigned long long* keys = small ? skeys : keys_g + (size_t)b * cap;   // cap is a power of two >= n (host guarantees)
    int* sup_g = ws + (size_t)b * cap;
    for (int i = tid; i < np2; i += NMS_T) {
        const unsigned long long k = i < n ? keys_g[(size_t)b * cap + i] : ~0ull;
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += NMS_T) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], c = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    // gather boxes in sorted order
    for (int i = tid; i < n; i += NMS_T) {
        const FaceRec& f = fc[(unsigned)(keys[i] & 0xffffffffu)];
        const int4 bx = make_int4(f.x, f.y, f.w, f.h);
        if (small) { sbox[i] = bx; ssup[i] = 0; }
        else sup_g[i] = 0;
    }
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        const bool dead = small ? ssup[i] != 0 : sup_g[i] != 0;
        if (!dead) {
            int4 bi;
            if (small) bi = sbox[i];
            else { const FaceRec& f = fc[(unsigned)(keys[i] & 0xffffffffu)]; bi = make_int4(f.x, f.y, f.w, f.h); }
            for (int j = i + 1 + tid; j < n; j += NMS_T) {
                if (small) {
                    if (!ssup[j] && iou_int(bi, sbox[j]) > thr) ssup[j] = 1;
                } else if (!sup_g[j]) {
                    const FaceRec& f = fc[(unsigned)(keys[j] & 0xffffffffu)];
                    if (iou_int(bi, make_int4(f.x, f.y, f.w, f.h)) > thr) sup_g[j] = 1;
                }
            }
        }
        __syncthreads();
    }
    // ordered compaction
    if (tid == 0) run_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += NMS_T) {
        const int i = base + tid;
        const bool keep = i < n && !(small ? ssup[i] != 0 : sup_g[i] != 0);
        const unsigned long long m = __ballot(keep);
        const int lane = tid & 63, wv = tid >> 6;
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = run_base;
        for (int w = 0; w < wv; ++w) off += wave_tot[w];
        if (keep && off + pre < max_out) out[(size_t)b * max_out + off + pre] = fc[(unsigned)(keys[i] & 0xffffffffu)];
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < NMS_T / 64; ++w) t += wave_tot[w]; run_base += t; }
        __syncthreads();
    }
    if (tid == 0) out_count[b] = run_base;
}

void launch_sort_nms(const FaceRec* cand, unsigned long long* keys, const int* count, int cap, int B, float nms_thr, FaceRec* out,
                     int* out_count, int max_out, int* order_ws, hipStream_t s) {
    hipLaunchKernelGGL(sort_nms_kernel, dim3(B), dim3(NMS_T), 0, s, cand, keys, count, cap, nms_thr, out, out_count, max_out,
                       order_ws);
}

// ------------------------------------------------------------------------------------------
// FaceRecognizer::alignFace (src/face_recognizer.cpp:93-133)
//   estimateAffinePartial2D (RANSAC-equivalent consensus over the 10 point pairs + least-squares
//   refit on the inliers, fp64) -> cv::warpAffine fixed-point bilinear, constant-0 border;
//   fallback when no transform exists: crop face.box & image and cv::resize to outW x outH.
// One workgroup per face.  ok: 1 = warped, 2 = fallback crop, 0 = empty result.
// ------------------------------------------------------------------------------------------
__device__ int estimate_similarity5(const float* from, const float* to, double* M) {
    int best_cnt = 0; unsigned best_mask = 0;
    for (int i = 0; i < 5; ++i)
        for (int j = i + 1; j < 5; ++j) {
            const double x1 = from[2 * i], y1 = from[2 * i + 1], x2 = from[2 * j], y2 = from[2 * j + 1];
            const double X1 = to[2 * i], Y1 = to[2 * i + 1], X2 = to[2 * j], Y2 = to[2 * j + 1];
            const double dx = x1 - x2, dy = y1 - y2;
            const double den = dx * dx + dy * dy;
            if (!(den > 0.0)) continue;
            const double dX = X1 - X2, dY = Y1 - Y2;
            const double a = (dX * dx + dY * dy) / den;
            const double b = (dY * dx - dX * dy) / den;
            const double tx = X1 - (a * x1 - b * y1);
            const double ty = Y1 - (b * x1 + a * y1);
            int cnt = 0; unsigned mask = 0;
            for (int p = 0; p < 5; ++p) {
                const double fx = from[2 * p], fy = from[2 * p + 1];
                const double ex = (a * fx - b * fy + tx) - to[2 * p];
                const double ey = (b * fx + a * fy + ty) - to[2 * p + 1];
                const double e = ex * ex + ey * ey;
                if (e <= 9.0) { ++cnt; mask |= 1u << p; }
            }
            // strictly more inliers, else the EARLIER pair stays (OpenCV's registrator only replaces on a larger count; an error-sum
            // tie-break would compare the rounding noise of a two-point consensus)
            if (cnt > best_cnt) { best_cnt = cnt; best_mask = mask; }
        }
    if (best_cnt < 2) return 0;
    double mx = 0, my = 0, mu = 0, mv = 0;
    for (int p = 0; p < 5; ++p)
        if (best_mask >> p & 1) { mx += from[2 * p]; my += from[2 * p + 1]; mu += to[2 * p]; mv += to[2 * p + 1]; }
    mx /= best_cnt; my /= best_cnt; mu /= best_cnt; mv /= best_cnt;
    double sxx = 0, sa = 0, sb = 0;
    for (int p = 0; p < 5; ++p)
        if (best_mask >> p & 1) {
            const double xc = from[2 * p] - mx, yc = from[2 * p + 1] - my;
            const double uc = to[2 * p] - mu, vc = to[2 * p + 1] - mv;
            sxx += xc * xc + yc * yc;
            sa += xc * uc + yc * vc;
            sb += xc * vc - yc * uc;
        }
    if (!(sxx > 0.0)) return 0;
    const double a = sa / sxx, b = sb / sxx;
    M[0] = a; M[1] = -b; M[2] = mu - (a * mx - b * my);
    M[3] = b; M[4] = a;  M[5] = mv - (b * mx + a * my);
    return 1;
}

constexpr int ALIGN_PARTS = 4;

__global__ __launch_bounds__(256) void align_kernel(const uint8_t* __restrict__ frames, long img_stride, int rows, int cols, int step,
                                                    const FaceRec* __restrict__ faces, const int* __restrict__ frame_of, int outH,
                                                    int outW, uint8_t* __restrict__ crops, int* __restrict__ ok) {
    __shared__ double Ms[6];
    __shared__ int mode;          // 1 warp, 2 crop-resize, 0 empty
    __shared__ int cbox[4];
    // ALIGN_PARTS workgroups per face: each repeats the (serial, fp64) transform estimate and warps its share of the pixels —
    // with one workgroup per face a 128-face batch left half the CUs idle behind that serial section
    const int n = blockIdx.x / ALIGN_PARTS, part = blockIdx.x - n * ALIGN_PARTS, tid = threadIdx.x;
    const FaceRec face = faces[n];
    const uint8_t* img = frames + (size_t)(frame_of ? frame_of[n] : n) * img_stride;
    if (tid == 0) {
        const float tmpl[10] = {38.2946f, 51.6963f, 73.5318f, 51.5014f, 56.0252f, 71.7366f, 41.5493f, 92.3655f, 70.7299f, 92.2041f};
        double M[6];
        if (estimate_similarity5(face.lm, tmpl, M)) {
            double D = M[0] * M[4] - M[1] * M[3];
            D = D != 0 ? 1. / D : 0;
            const double A11 = M[4] * D, A22 = M[0] * D;
            M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
            const double b1 = -M[0] * M[2] - M[1] * M[5];
            const double b2 = -M[3] * M[2] - M[4] * M[5];
            M[2] = b1; M[5] = b2;
            for (int i = 0; i < 6; ++i) Ms[i] = M[i];
            mode = 1;
        } else {
            const int x0 = max(face.x, 0), y0 = max(face.y, 0);
            const int x1 = min(face.x + face.w, cols), y1 = min(face.y + face.h, rows);
            if (x1 - x0 > 0 && y1 - y0 > 0) { cbox[0] = x0; cbox[1] = y0; cbox[2] = x1 - x0; cbox[3] = y1 - y0; mode = 2; }
            else mode = 0;
        }
        if (part == 0) ok[n] = mode;
    }
    __syncthreads();
    uint8_t* dst = crops + (size_t)n * outH * outW * 3;
    const int npx = outH * outW;
    const int px_lo = (int)((long)npx * part / ALIGN_PARTS), px_hi = (int)((long)npx * (part + 1) / ALIGN_PARTS);
    if (mode == 0) {
        for (int i = px_lo * 3 + tid; i < px_hi * 3; i += 256) dst[i] = 0;
        return;
    }
    if (mode == 2) {
        const uint8_t* s = img + (size_t)cbox[1] * step + cbox[0] * 3;
        for (int i = px_lo + tid; i < px_hi; i += 256) {
            const int x = i % outW, y = i / outW;
            for (int c = 0; c < 3; ++c) dst[i * 3 + c] = (uint8_t)resize_px(s, cbox[3], cbox[2], step, outH, outW, x, y, c);
        }
        return;
    }
    const double m0 = Ms[0], m1 = Ms[1], m2 = Ms[2], m3 = Ms[3], m4 = Ms[4], m5 = Ms[5];
    for (int i = px_lo + tid; i < px_hi; i += 256) {
        const int x = i % outW, y = i / outW;
        const int X0 = cv_round_d((m1 * y + m2) * 1024) + 16;
        const int Y0 = cv_round_d((m4 * y + m5) * 1024) + 16;
        const int adelta = cv_round_d(m0 * x * 1024);
        const int bdelta = cv_round_d(m3 * x * 1024);
        const int X = (int)((unsigned)X0 + (unsigned)adelta) >> 5;
        const int Y = (int)((unsigned)Y0 + (unsigned)bdelta) >> 5;
        const int ix = sat_short(X >> 5), iy = sat_short(Y >> 5);
        const int fx = X & 31, fy = Y & 31;
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        const bool r0 = iy >= 0 && iy < rows, r1 = iy + 1 >= 0 && iy + 1 < rows;
        const bool c0 = ix >= 0 && ix < cols, c1 = ix + 1 >= 0 && ix + 1 < cols;
        const uint8_t* p0 = img + (size_t)iy * step + ix * 3;
        const uint8_t* p1 = p0 + step;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p00 = (r0 && c0) ? p0[c] : 0, p01 = (r0 && c1) ? p0[3 + c] : 0;
            const int p10 = (r1 && c0) ? p1[c] : 0, p11 = (r1 && c1) ? p1[3 + c] : 0;
            const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
            dst[i * 3 + c] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

void launch_align(const uint8_t* frames, long img_stride, int rows, int cols, int step, const FaceRec* faces, const int* frame_of,
                  int n, int outH, int outW, uint8_t* crops, int* ok, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(align_kernel, dim3(n * ALIGN_PARTS), dim3(256), 0, s, frames, img_stride, rows, cols, step, faces, frame_of, outH, outW, crops, ok);
}

// ------------------------------------------------------------------------------------------
// FaceRecognizer::normalize (src/face_recognizer.cpp:306-318): v / sqrt(sum v^2) when > 0.
// One wave per embedding; the sum is a lane-strided partial + butterfly reduction (differs from
// the reference's sequential fp32 sum by rounding only).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int dim) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* v = in + (size_t)row * dim;
    float s = 0.f;
    for (int i = lane; i < dim; i += 64) s += v[i] * v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float norm = sqrtf(s);
    for (int i = lane; i < dim; i += 64) out[(size_t)row * dim + i] = norm > 0.f ? v[i] / norm : v[i];
}

void launch_l2_normalize(const float* in, float* out, int n, int dim, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(l2norm_kernel, dim3((n + 3) / 4), dim3(256), 0, s, in, out, n, dim);
}

// ------------------------------------------------------------------------------------------
// FaceRecognizer::compareFaces generalised to a gallery (src/face_recognizer.cpp:320-334):
// mapped score (dot + 1) / 2, ranked (score desc, gallery index asc).  The scan itself is
// gallery.hip (one pass, per-workgroup top-k lists); here: the merge of those lists — also the
// merge step of a row-sharded gallery (fh_topk_merge_dev).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(float s1, int i1, float s2, int i2) { return s1 > s2 || (s1 == s2 && i1 < i2); }

// one workgroup per query: k rounds of "best entry that comes after the previous pick".
// CACHED: the nparts * k <= 8192 candidate entries are read ONCE into registers (32 per thread) and every round is a register scan +
// a wave reduction + one LDS hand-off between the four waves; otherwise each round re-reads the lists from memory (L2).
template <bool CACHED>
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ ps, const int* __restrict__ pi, int nparts, int Q, int k,
                                                         long part_stride, float* __restrict__ out_s, int* __restrict__ out_i) {
    __shared__ float rs[256];
    __shared__ int ri[256];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int total = nparts * k;
    constexpr int E = 32;
    float es[E]; int ei[E];
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const int e = j * 256 + tid;
            es[j] = -INFINITY; ei[j] = -1;
            if (e < total) {
                const int part = e / k, pos = e - part * k;
                const size_t o = (size_t)part * part_stride + (size_t)q * k + pos;
                es[j] = ps[o]; ei[j] = pi[o];
            }
        }
    }
    float last_s = 0.f; int last_i = -1; bool have_last = false, exhausted = false;
    for (int round = 0; round < k; ++round) {
        float best_s = -INFINITY; int best_i = 0x7fffffff;                   // (emptiness is told by the index, not by the score: rows need not be unit vectors)
        if (CACHED) {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const float sc = es[j]; const int gi = ei[j];
                const bool ok = !exhausted && gi >= 0 && (!have_last || better(last_s, last_i, sc, gi)) && better(sc, gi, best_s, best_i);
                best_s = ok ? sc : best_s; best_i = ok ? gi : best_i;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {               // wave reduction
                const float os = __shfl_xor(best_s, o); const int oi = __shfl_xor(best_i, o);
                const bool t = better(os, oi, best_s, best_i);
                best_s = t ? os : best_s; best_i = t ? oi : best_i;
            }
            if (lane == 0) { rs[wv] = best_s; ri[wv] = best_i; }
            __syncthreads();
            best_s = rs[0]; best_i = ri[0];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (better(rs[w], ri[w], best_s, best_i)) { best_s = rs[w]; best_i = ri[w]; }
            __syncthreads();
            last_s = best_s; last_i = best_i;
        } else {
            for (int e = tid; e < total; e += 256) {
                const int part = e / k, pos = e - part * k;
                const size_t o = (size_t)part * part_stride + (size_t)q * k + pos;
                const float sc = ps[o]; const int gi = pi[o];
                if (gi < 0 || exhausted) continue;
                if (have_last && !better(last_s, last_i, sc, gi)) continue;     // must come strictly after the last pick
                if (better(sc, gi, best_s, best_i)) { best_s = sc; best_i = gi; }
            }
            rs[tid] = best_s; ri[tid] = best_i;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if (tid < st && better(rs[tid + st], ri[tid + st], rs[tid], ri[tid])) { rs[tid] = rs[tid + st]; ri[tid] = ri[tid + st]; }
                __syncthreads();
            }
            last_s = rs[0]; last_i = ri[0];
            __syncthreads();
        }
        have_last = true;
        if (tid == 0) {
            const bool found = last_i != 0x7fffffff;
            out_s[(size_t)q * k + round] = found ? last_s : -1.0f;
            out_i[(size_t)q * k + round] = found ? last_i : -1;
        }
        if (last_i == 0x7fffffff) exhausted = true;      // nothing left: later rounds find nothing either (workgroup-uniform)
    }
}

// Match / Unknown decision of the reference's webcam loop (src/main.cpp:229-233): a query is labelled with its best
// gallery row when the mapped score (dot+1)/2 is STRICTLY above the threshold, else -1.
__global__ void label_kernel(const float* __restrict__ best_s, const int* __restrict__ best_i, int n, float thr, int* __restrict__ labels) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) labels[q] = (best_i[q] >= 0 && best_s[q] > thr) ? best_i[q] : -1;
}
void launch_label(const float* best_score, const int* best_idx, int n, float thr, int* labels, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(label_kernel, dim3((n + 255) / 256), dim3(256), 0, s, best_score, best_idx, n, thr, labels);
}

// part_stride = words between the lists of consecutive parts (Q * k when they are packed; the sharded exchange of comm.cpp interleaves
// score and index planes per rank)
void launch_topk_merge_strided(const float* part_score, const int* part_idx, int nparts, int Q, int k, long part_stride, float* out_score,
                               int* out_idx, hipStream_t s) {
    if ((long)nparts * k <= 8192) hipLaunchKernelGGL(topk_merge_kernel<true>, dim3(Q), dim3(256), 0, s, part_score, part_idx, nparts, Q, k, part_stride, out_score, out_idx);
    else hipLaunchKernelGGL(topk_merge_kernel<false>, dim3(Q), dim3(256), 0, s, part_score, part_idx, nparts, Q, k, part_stride, out_score, out_idx);
}
void launch_topk_merge(const float* part_score, const int* part_idx, int nparts, int Q, int k, float* out_score, int* out_idx,
                       hipStream_t s) {
    launch_topk_merge_strided(part_score, part_idx, nparts, Q, k, (long)Q * k, out_score, out_idx, s);
}

// ------------------------------------------------------------------------------------------
// Hand-off between detect and embed without a host round trip of the boxes: take the first
// min(count, F) faces of every frame (they are score-ordered, as main.cpp:101-104 relies on)
// and write them densely, with the frame index of each.  total[0] = number of faces.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_faces_kernel(const FaceRec* __restrict__ det, const int* __restrict__ counts, int n,
                                                           int per_frame, int F, FaceRec* __restrict__ faces, int* __restrict__ frame_of,
                                                           int* __restrict__ total) {
    __shared__ int offs[4097];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < n; ++b) { offs[b] = acc; acc += min(min(counts[b], per_frame), F); }
        offs[n] = acc;
        total[0] = acc;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n; b += 256) {
        const int c = offs[b + 1] - offs[b];
        for (int j = 0; j < c; ++j) { faces[offs[b] + j] = det[(size_t)b * per_frame + j]; frame_of[offs[b] + j] = b; }
    }
}

void launch_select_faces(const FaceRec* det, const int* counts, int n, int per_frame, int F, FaceRec* faces, int* frame_of, int* total,
                         hipStream_t s) {
    hipLaunchKernelGGL(select_faces_kernel, dim3(1), dim3(256), 0, s, det, counts, n, per_frame, F, faces, frame_of, total);
}

}  // namespace fh
