/* face_oracle.c — CPU ORACLE (test infrastructure only; never shipped, never timed as the product).
 *
 * Plain-C fp32 restatement of the reference hot path
 *   FaceDetector::detect()  ->  FaceRecognizer::extractFeature()  ->  compareFaces()
 * of cucibala/FaceRecognizeOnnx.  Every function cites the reference lines it follows
 * (paths relative to the reference root).  The arithmetic the reference delegates to
 * ONNX Runtime (graph execution) and OpenCV (resize / estimateAffinePartial2D / warpAffine)
 * is restated from their published algorithms; neither library
 * nor any model file is available offline and the reference holds no golden vectors, so
 *
 *      PARITY WITH ORT / OPENCV IS UNPINNED.
 *
 * The restatement is instead pinned by (i) hand-derived known-answer tests and (ii) an
 * independent PyTorch-CPU fp64 evaluation of the same graphs (tests/, tests/golden/).
 *
 * Layout convention here is the reference's: NCHW fp32 activations, HWC BGR u8 images.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* POD mirror of reference `struct FaceBox` (src/face_detector.h:8-12):
 * cv::Rect{x,y,width,height} + float score + cv::Point2f landmarks[5]  = 60 bytes. */
typedef struct {
    int32_t x, y, w, h;
    float score;
    float lm[10];
} orc_face;

ORC_API void orc_set_threads(int n) {
#if defined(_OPENMP)
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Graph operators (what `session_->Run` computes: src/face_detector.cpp:179-183,
 * src/face_recognizer.cpp:279-283).  Unfused, NCHW, fp32 accumulate, K order (ci,ky,kx).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_conv2d(const float* x, int N, int Cin, int H, int W,
                        const float* w, const float* bias, int Cout,
                        int kh, int kw, int stride, int pad, int group, float* y) {
    const int Ho = (H + 2 * pad - kh) / stride + 1;
    const int Wo = (W + 2 * pad - kw) / stride + 1;
    const int cin_g = Cin / group, cout_g = Cout / group;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n) {
        for (int co = 0; co < Cout; ++co) {
            float* yp = y + ((size_t)n * Cout + co) * Ho * Wo;
            const float b = bias ? bias[co] : 0.0f;
            for (int i = 0; i < Ho * Wo; ++i) yp[i] = b;
            const int g = co / cout_g;
            for (int cig = 0; cig < cin_g; ++cig) {
                const int ci = g * cin_g + cig;
                const float* xp = x + ((size_t)n * Cin + ci) * H * W;
                const float* wp = w + (((size_t)co * cin_g + cig) * kh) * kw;
                for (int ky = 0; ky < kh; ++ky) {
                    for (int kx = 0; kx < kw; ++kx) {
                        const float wv = wp[ky * kw + kx];
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * stride + ky - pad;
                            if (iy < 0 || iy >= H) continue;
                            /* ox range with 0 <= ox*stride + kx - pad < W */
                            int ox0 = 0, ox1 = Wo;
                            while (ox0 < Wo && ox0 * stride + kx - pad < 0) ++ox0;
                            while (ox1 > ox0 && (ox1 - 1) * stride + kx - pad >= W) --ox1;
                            const float* xr = xp + (size_t)iy * W + (kx - pad);
                            float* yr = yp + (size_t)oy * Wo;
                            if (stride == 1) {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox];
                            } else {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox * stride];
                            }
                        }
                    }
                }
            }
        }
    }
}

/* ONNX BatchNormalization (inference): y = (x - mean) / sqrt(var + eps) * gamma + beta. */
ORC_API void orc_batchnorm(const float* x, int N, int C, int HW, const float* gamma,
                           const float* beta, const float* mean, const float* var,
                           float eps, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float inv = 1.0f / sqrtf(var[c] + eps);
            const float s = gamma[c] * inv;
            const float t = beta[c] - mean[c] * s;
            const float* xp = x + ((size_t)n * C + c) * HW;
            float* yp = y + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) yp[i] = xp[i] * s + t;
        }
}

ORC_API void orc_prelu(const float* x, int N, int C, int HW, const float* slope, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float s = slope[c];
            const float* xp = x + ((size_t)n * C + c) * HW;
            float* yp = y + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i) yp[i] = xp[i] >= 0.0f ? xp[i] : xp[i] * s;
        }
}

ORC_API void orc_relu(const float* x, size_t n, float* y) {
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.0f ? x[i] : 0.0f;
}

ORC_API void orc_sigmoid(const float* x, size_t n, float* y) {
    for (size_t i = 0; i < n; ++i) y[i] = 1.0f / (1.0f + expf(-x[i]));
}

ORC_API void orc_add(const float* a, const float* b, size_t n, float* y) {
    for (size_t i = 0; i < n; ++i) y[i] = a[i] + b[i];
}

/* ONNX Resize, mode=nearest, integer scale (asymmetric / floor): y[oy][ox] = x[oy/s][ox/s]. */
ORC_API void orc_resize_nearest(const float* x, int NC, int H, int W, int s, float* y) {
    const int Ho = H * s, Wo = W * s;
    for (int p = 0; p < NC; ++p)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox)
                y[((size_t)p * Ho + oy) * Wo + ox] = x[((size_t)p * H + oy / s) * W + ox / s];
}

/* ONNX Transpose perm=[0,2,3,1]. */
ORC_API void orc_nchw_to_nhwc(const float* x, int N, int C, int H, int W, float* y) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < H * W; ++i)
                y[((size_t)n * H * W + i) * C + c] = x[((size_t)n * C + c) * H * W + i];
}

/* ONNX Gemm with transB=1, alpha=beta=1: y[M,N] = x[M,K] * w[N,K]^T + b[N]. */
ORC_API void orc_gemm_nt(const float* x, int M, int K, const float* w, const float* b, int N, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const float* xp = x + (size_t)m * K;
            const float* wp = w + (size_t)n * K;
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc += xp[k] * wp[k];
            y[(size_t)m * N + n] = acc + (b ? b[n] : 0.0f);
        }
}

/* ------------------------------------------------------------------------------------------
 * OpenCV pieces used by the reference, restated ([EXT]).
 * ---------------------------------------------------------------------------------------- */
static inline int cv_round(double v) { return (int)lrint(v); }          /* cvRound: half-to-even */
static inline int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static inline int sat_int(double v) {
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return cv_round(v);
}
static inline short sat_short_i(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline short sat_short_f(float v) { return sat_short_i(cv_round(v)); }

/* cv::resize(src, dst, Size(dw,dh)) with the default INTER_LINEAR on CV_8UC3
 * (reference call sites src/face_detector.cpp:117, src/face_recognizer.cpp:123,170).
 * Classic fixed-point path (11-bit coefficients); same size = copy; exact 2x shrink is
 * dispatched to INTER_AREA (2x2 mean) as OpenCV does. */
ORC_API void orc_resize_bilinear_u8c3(const uint8_t* src, int sh, int sw, int sstep,
                                      uint8_t* dst, int dh, int dw, int dstep) {
    if (dh == sh && dw == sw) {
        for (int y = 0; y < sh; ++y) memcpy(dst + (size_t)y * dstep, src + (size_t)y * sstep, (size_t)sw * 3);
        return;
    }
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1.0 / inv_sx, scale_y = 1.0 / inv_sy;
    const int isx = sat_int(scale_x), isy = sat_int(scale_y);
    const int area_fast = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    if (area_fast && isx == 2 && isy == 2) {
        for (int y = 0; y < dh; ++y) {
            const uint8_t* s0 = src + (size_t)(2 * y) * sstep;
            const uint8_t* s1 = s0 + sstep;
            uint8_t* d = dst + (size_t)y * dstep;
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < 3; ++c)
                    d[x * 3 + c] = (uint8_t)((s0[6 * x + c] + s0[6 * x + 3 + c] + s1[6 * x + c] + s1[6 * x + 3 + c] + 2) >> 2);
        }
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short_f((1.f - fx) * 2048.f);
        ialpha[2 * dx + 1] = sat_short_f(fx * 2048.f);
    }
    int* rows = (int*)malloc(sizeof(int) * 2 * dw * 3);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        const short b0 = sat_short_f((1.f - fy) * 2048.f), b1 = sat_short_f(fy * 2048.f);
        for (int k = 0; k < 2; ++k) {
            int yy = sy + k;
            yy = yy >= 0 ? (yy < sh ? yy : sh - 1) : 0;
            const uint8_t* S = src + (size_t)yy * sstep;
            int* D = rows + (size_t)k * dw * 3;
            for (int dx = 0; dx < dw; ++dx) {
                const int sx = xofs[dx] * 3;
                for (int c = 0; c < 3; ++c)
                    D[dx * 3 + c] = dx < xmax ? S[sx + c] * ialpha[2 * dx] + S[sx + 3 + c] * ialpha[2 * dx + 1]
                                              : S[sx + c] * 2048;
            }
        }
        uint8_t* d = dst + (size_t)dy * dstep;
        const int* S0 = rows;
        const int* S1 = rows + (size_t)dw * 3;
        for (int i = 0; i < dw * 3; ++i) {
            int v = (((b0 * (S0[i] >> 4)) >> 16) + ((b1 * (S1[i] >> 4)) >> 16) + 2) >> 2;
            d[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(rows); free(ialpha); free(xofs);
}

/* cv::estimateAffinePartial2D(from, to) with its defaults (RANSAC, 3.0 px, refine) as the
 * reference calls it (src/face_recognizer.cpp:110-113), for exactly 5 point pairs.
 * OpenCV draws random 2-point samples from a fixed-seed RNG, keeps the model with the most
 * inliers (squared error <= 9) and refines it on those inliers; the 4-DoF model is linear,
 * so the refinement converges to the closed-form least squares over the inlier set.  With 5
 * points the sample space is the C(5,2)=10 pairs, which are enumerated exhaustively here:
 * best = most inliers, then FIRST pair in (i < j) order — OpenCV's registrator replaces its
 * best model only on a STRICTLY larger inlier count, so among equals the earliest sample stays.
 * (Rounds 1-2 broke such ties by the smaller inlier error sum; for a two-point consensus that
 * sum is rounding noise around 1e-25, and a 3e-5 px landmark difference chose another model:
 * found by the round-3 composition test.)  "RANSAC-equivalent, not RNG-identical" (SURVEY.md B.3).  M = [[a,-b,tx],[b,a,ty]] (double).  Returns 0 when
 * no pair of distinct source points exists (OpenCV returns an empty Mat). */
ORC_API int orc_estimate_similarity5(const float* from, const float* to, double* M) {
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
    M[0] = a;  M[1] = -b; M[2] = mu - (a * mx - b * my);
    M[3] = b;  M[4] = a;  M[5] = mv - (b * mx + a * my);
    return 1;
}

/* cv::warpAffine(src, dst, M, Size(dw,dh)) defaults: INTER_LINEAR, BORDER_CONSTANT(0), M is
 * the forward map and is inverted first (reference src/face_recognizer.cpp:129-130).
 * Classic fixed-point path: coordinates in 1/1024 px rounded to int, +16, >>5 -> 1/32 px;
 * bilinear weights (32-fx)(32-fy)*32 etc. (sum 32768), result (sum + 2^14) >> 15; taps that
 * fall outside the image read 0.  CV_8UC3 only. */
ORC_API void orc_warp_affine_u8c3(const uint8_t* src, int sh, int sw, int sstep,
                                  const double* Mfwd, uint8_t* dst, int dh, int dw, int dstep) {
    double M[6];
    memcpy(M, Mfwd, sizeof(M));
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    for (int y = 0; y < dh; ++y) {
        const int X0 = sat_int((M[1] * y + M[2]) * 1024) + 16;
        const int Y0 = sat_int((M[4] * y + M[5]) * 1024) + 16;
        uint8_t* d = dst + (size_t)y * dstep;
        for (int x = 0; x < dw; ++x) {
            const int adelta = sat_int(M[0] * x * 1024);
            const int bdelta = sat_int(M[3] * x * 1024);
            const int X = (int)((unsigned)X0 + (unsigned)adelta) >> 5;   /* wraps like OpenCV's int add */
            const int Y = (int)((unsigned)Y0 + (unsigned)bdelta) >> 5;
            const int ix = sat_short_i(X >> 5), iy = sat_short_i(Y >> 5);
            const int fx = X & 31, fy = Y & 31;
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
            const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            for (int c = 0; c < 3; ++c) {
                int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
                if (iy >= 0 && iy < sh) {
                    if (ix >= 0 && ix < sw) p00 = src[(size_t)iy * sstep + ix * 3 + c];
                    if (ix + 1 >= 0 && ix + 1 < sw) p01 = src[(size_t)iy * sstep + (ix + 1) * 3 + c];
                }
                if (iy + 1 >= 0 && iy + 1 < sh) {
                    if (ix >= 0 && ix < sw) p10 = src[(size_t)(iy + 1) * sstep + ix * 3 + c];
                    if (ix + 1 >= 0 && ix + 1 < sw) p11 = src[(size_t)(iy + 1) * sstep + (ix + 1) * 3 + c];
                }
                const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
                d[x * 3 + c] = (uint8_t)(v > 255 ? 255 : v);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * FaceDetector::preprocess  (src/face_detector.cpp:92-137)
 * Returns 0 on the reference's failure paths (inputData left empty, scale = 1).
 * ---------------------------------------------------------------------------------------- */
ORC_API int orc_det_preprocess(const uint8_t* bgr, int rows, int cols, int step,
                               int inW, int inH, float* out /*[3,inH,inW]*/, float* scale_out) {
    if (!bgr || cols <= 0 || rows <= 0) { *scale_out = 1.0f; return 0; }          /* :94-98 */
    const float scaleW = (float)inW / cols;                                          /* :101 */
    const float scaleH = (float)inH / rows;                                          /* :102 */
    const float scale = scaleW < scaleH ? scaleW : scaleH;                           /* :103 */
    const int newW = (int)(cols * scale);                                            /* :105 */
    const int newH = (int)(rows * scale);                                            /* :106 */
    if (newW <= 0 || newH <= 0) { *scale_out = 1.0f; return 0; }                     /* :109-113 */
    *scale_out = scale;
    uint8_t* padded = (uint8_t*)calloc((size_t)inH * inW * 3, 1);                    /* :120 zeros */
    /* :117 resize, :121 paste top-left.  (Reference would throw cv::Exception if the ROI
       exceeded the canvas; with scale=min(...) it never does.) */
    orc_resize_bilinear_u8c3(bgr, rows, cols, step, padded, newH, newW, inW * 3);
    for (int c = 0; c < 3; ++c)                                                      /* :129 */
        for (int h = 0; h < inH; ++h)
            for (int w = 0; w < inW; ++w)                                            /* :125 BGR->RGB */
                out[((size_t)c * inH + h) * inW + w] =
                    (padded[((size_t)h * inW + w) * 3 + (2 - c)] - 127.5f) / 128.0f;  /* :133 */
    free(padded);
    return 1;
}

/* Canonical SCRFD anchor decode (InsightFace model_zoo/scrfd.py) — the step
 * the reference lacks.  Produces the [N,15] rows the reference's
 * postprocess consumes: x1,y1,x2,y2,score,kx0,ky0..kx4,ky4.  Row order: stride 8 rows,
 * then 16, then 32; inside a stride r = (gy*(W/s)+gx)*2 + a. */
ORC_API int orc_scrfd_decode(const float* const* score, const float* const* bbox,
                             const float* const* kps, int inH, int inW, float* rows15) {
    static const int strides[3] = {8, 16, 32};
    int r = 0;
    for (int si = 0; si < 3; ++si) {
        const int s = strides[si], gh = inH / s, gw = inW / s;
        for (int gy = 0; gy < gh; ++gy)
            for (int gx = 0; gx < gw; ++gx)
                for (int a = 0; a < 2; ++a) {
                    const int i = (gy * gw + gx) * 2 + a;
                    const float cx = (float)(gx * s), cy = (float)(gy * s);
                    const float* d = bbox[si] + (size_t)i * 4;
                    const float* k = kps[si] + (size_t)i * 10;
                    float* o = rows15 + (size_t)r * 15;
                    o[0] = cx - d[0] * (float)s;
                    o[1] = cy - d[1] * (float)s;
                    o[2] = cx + d[2] * (float)s;
                    o[3] = cy + d[3] * (float)s;
                    o[4] = score[si][i];
                    for (int j = 0; j < 5; ++j) {
                        o[5 + 2 * j] = cx + k[2 * j] * (float)s;
                        o[6 + 2 * j] = cy + k[2 * j + 1] * (float)s;
                    }
                    ++r;
                }
    }
    return r;
}

/* FaceDetector::iou  (src/face_detector.cpp:340-354) — integer intersection / integer
 * denominator, one float divide; 0/0 = NaN compares false against the threshold. */
ORC_API float orc_iou(const orc_face* a, const orc_face* b) {
    const int x1 = a->x > b->x ? a->x : b->x;
    const int y1 = a->y > b->y ? a->y : b->y;
    const int ax2 = a->x + a->w, bx2 = b->x + b->w, ay2 = a->y + a->h, by2 = b->y + b->h;
    const int x2 = ax2 < bx2 ? ax2 : bx2;
    const int y2 = ay2 < by2 ? ay2 : by2;
    const int w = x2 - x1 > 0 ? x2 - x1 : 0;
    const int h = y2 - y1 > 0 ? y2 - y1 : 0;
    const int inter = w * h;
    const int area1 = a->w * a->h, area2 = b->w * b->h;
    return (float)inter / (float)(area1 + area2 - inter);
}

/* FaceDetector::nms  (src/face_detector.cpp:356-384).  std::sort there is unstable, so tie
 * order is unspecified; this build fixes the total order (score desc, candidate index asc). */
typedef struct { float score; int idx; } orc_key;
static int key_cmp(const void* pa, const void* pb) {
    const orc_key* a = (const orc_key*)pa; const orc_key* b = (const orc_key*)pb;
    if (a->score > b->score) return -1;
    if (a->score < b->score) return 1;
    return a->idx < b->idx ? -1 : a->idx > b->idx;
}
ORC_API int orc_nms(orc_face* boxes, int n, float thr) {
    if (n <= 0) return 0;
    orc_key* keys = (orc_key*)malloc(sizeof(orc_key) * n);
    orc_face* sorted = (orc_face*)malloc(sizeof(orc_face) * n);
    char* sup = (char*)calloc(n, 1);
    for (int i = 0; i < n; ++i) { keys[i].score = boxes[i].score; keys[i].idx = i; }
    qsort(keys, n, sizeof(orc_key), key_cmp);                                /* :357-359 */
    for (int i = 0; i < n; ++i) sorted[i] = boxes[keys[i].idx];
    for (int i = 0; i < n; ++i) {                                            /* :363-374 */
        if (sup[i]) continue;
        for (int j = i + 1; j < n; ++j) {
            if (sup[j]) continue;
            if (orc_iou(&sorted[i], &sorted[j]) > thr) sup[j] = 1;           /* strict > :370 */
        }
    }
    int m = 0;
    for (int i = 0; i < n; ++i) if (!sup[i]) boxes[m++] = sorted[i];         /* :376-383 */
    free(sup); free(sorted); free(keys);
    return m;
}

/* FaceDetector::postprocess row loop  (src/face_detector.cpp:249-278 / :286-325):
 * strict score > thr, /scale, int truncation, width from the float difference.
 * Returns the number of boxes BEFORE nms (call orc_nms next, as :333 does). */
ORC_API int orc_postprocess_rows(const float* rows, int n, int feat, float scale, float thr,
                                 orc_face* out, int max_out) {
    int m = 0;
    if (feat < 15) return 0;                                                 /* :300-303 */
    for (int i = 0; i < n; ++i) {
        const float* o = rows + (size_t)i * feat;
        const float score = o[4];
        if (score > thr) {                                                   /* :253 / :305 */
            if (m >= max_out) break;
            const float x1 = o[0] / scale, y1 = o[1] / scale, x2 = o[2] / scale, y2 = o[3] / scale;
            orc_face f;
            f.x = (int)x1; f.y = (int)y1; f.w = (int)(x2 - x1); f.h = (int)(y2 - y1);   /* :260-265 */
            f.score = score;
            for (int j = 0; j < 5; ++j) {                                    /* :270-273 */
                f.lm[2 * j] = o[5 + 2 * j] / scale;
                f.lm[2 * j + 1] = o[6 + 2 * j] / scale;
            }
            out[m++] = f;
        }
    }
    return m;
}

/* ------------------------------------------------------------------------------------------
 * FaceRecognizer::alignFace  (src/face_recognizer.cpp:93-133)
 * Returns 1 and fills out112 (112x112x3 BGR u8) or 0 for the "empty Mat" results.
 * ---------------------------------------------------------------------------------------- */
static const float k_template[10] = {38.2946f, 51.6963f, 73.5318f, 51.5014f, 56.0252f,
                                     71.7366f, 41.5493f, 92.3655f, 70.7299f, 92.2041f}; /* :101-107 */

ORC_API int orc_align_face(const uint8_t* bgr, int rows, int cols, int step,
                           const orc_face* face, int outW, int outH, uint8_t* out) {
    if (!bgr || rows <= 0 || cols <= 0) return 0;                            /* :95-98 */
    double M[6];
    if (!orc_estimate_similarity5(face->lm, k_template, M)) {                /* :110-116 */
        /* fallback: crop face.box & image, cv::resize to the input size  :119-126 */
        int x0 = face->x > 0 ? face->x : 0, y0 = face->y > 0 ? face->y : 0;
        int x1 = face->x + face->w < cols ? face->x + face->w : cols;
        int y1 = face->y + face->h < rows ? face->y + face->h : rows;
        if (x1 - x0 > 0 && y1 - y0 > 0) {
            orc_resize_bilinear_u8c3(bgr + (size_t)y0 * step + x0 * 3, y1 - y0, x1 - x0, step,
                                     out, outH, outW, outW * 3);
            return 1;
        }
        return 0;
    }
    orc_warp_affine_u8c3(bgr, rows, cols, step, M, out, outH, outW, outW * 3);   /* :130 */
    return 1;
}

/* FaceRecognizer::preprocess  (src/face_recognizer.cpp:135-150) */
ORC_API void orc_rec_preprocess(const uint8_t* bgr, int H, int W, float* out /*[3,H,W]*/) {
    for (int c = 0; c < 3; ++c)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                out[((size_t)c * H + h) * W + w] = (bgr[((size_t)h * W + w) * 3 + (2 - c)] - 127.5f) / 128.0f;
}

/* FaceRecognizer::normalize  (src/face_recognizer.cpp:306-318): sequential fp32 sum. */
ORC_API void orc_l2_normalize(float* v, int n) {
    float norm = 0.0f;
    for (int i = 0; i < n; ++i) norm += v[i] * v[i];
    norm = sqrtf(norm);
    if (norm > 0) for (int i = 0; i < n; ++i) v[i] /= norm;
}

/* FaceRecognizer::compareFaces  (src/face_recognizer.cpp:320-334) */
ORC_API float orc_compare(const float* a, int na, const float* b, int nb) {
    if (na != nb || na == 0) return 0.0f;
    float dot = 0.0f;
    for (int i = 0; i < na; ++i) dot += a[i] * b[i];
    return (dot + 1.0f) / 2.0f;
}

/* 1:N generalisation of compareFaces: mapped score of q against every
 * gallery row, top-k by (score desc, index asc). */
ORC_API void orc_gallery_topk(const float* q, int Q, const float* gal, int G, int dim, int k,
                              float* out_score, int* out_idx) {
#pragma omp parallel for schedule(static)
    for (int qi = 0; qi < Q; ++qi) {
        float* bs = out_score + (size_t)qi * k; int* bi = out_idx + (size_t)qi * k;
        int cnt = 0;
        for (int g = 0; g < G; ++g) {
            const float s = orc_compare(q + (size_t)qi * dim, dim, gal + (size_t)g * dim, dim);
            if (cnt < k || s > bs[cnt - 1]) {
                int p = cnt < k ? cnt++ : k - 1;
                while (p > 0 && bs[p - 1] < s) { bs[p] = bs[p - 1]; bi[p] = bi[p - 1]; --p; }
                bs[p] = s; bi[p] = g;
            }
        }
        for (int p = cnt; p < k; ++p) { bs[p] = -1.0f; bi[p] = -1; }
    }
}
