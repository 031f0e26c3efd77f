// image_io.cpp — file -> BGR u8 image, the step in front of the hot path.
//
// What it replaces: `cv::imread(path)` in the reference's callers (src/main.cpp:42,71-72,140-141), i.e.
// OpenCV's default IMREAD_COLOR decode: 8-bit, 3 channels, B-G-R order, alpha dropped, grey replicated,
// 16-bit samples reduced to their high byte, EXIF orientation applied to JPEGs.  OpenCV delegates JPEG to
// libjpeg(-turbo) with its defaults (integer "islow" IDCT, "fancy" triangle chroma up-sampling, 16-bit
// fixed-point YCbCr->RGB); those three published algorithms are restated here so that the pixels are the
// ones the reference's detector would have seen.  Pinned by tests/test_image_io.py against Pillow (which
// wraps the same libjpeg-turbo / zlib) on files written by tests/golden/make_images.py.
//
// Formats: JPEG (baseline, extended-sequential and progressive Huffman, 8-bit, grey / YCbCr / RGB, any
// sampling factors libjpeg up-samples with h2v1 / h2v2 / h1v2 "fancy" or integer replication, restart
// intervals), PNG (all colour types, 1..16 bit, Adam7; inflate from zlib), BMP (24/32-bit BI_RGB),
// binary PPM / PGM.  Host code: runs on the CPU exactly as the reference's imread does.
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/facehip.h"

namespace fh {
void set_error(const std::string& msg);      // api.cpp
}

namespace {

struct Image {
    int rows = 0, cols = 0;
    std::vector<uint8_t> bgr;
};

struct DecodeError {
    std::string msg;
};
[[noreturn]] void fail(const std::string& m) { throw DecodeError{m}; }

// Header sanity: a corrupt file must not make the decoder allocate gigabytes (OpenCV: CV_IO_MAX_IMAGE_PIXELS = 2^30).
constexpr long long kMaxPixels = 1LL << 28;
void check_size(long long w, long long h, const char* fmt) {
    if (w <= 0 || h <= 0 || w > (1 << 20) || h > (1 << 20) || w * h > kMaxPixels) fail(std::string(fmt) + ": unreasonable image size");
}

// =====================================================================================================
// JPEG
// =====================================================================================================
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    // canonical decode tables (ITU T.81 F.2.2.3)
    int mincode[17], maxcode[18], valptr[17];
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int wblocks = 0, hblocks = 0;            // allocated block grid (padded to whole MCUs)
    int dw = 0, dh = 0;                      // "downsampled" real sample dimensions
    std::vector<int16_t> coef;               // [hblocks][wblocks][64], natural order
    uint16_t quant[64];
    bool quant_latched = false;
    std::vector<uint8_t> plane;              // IDCT output, (hblocks*8) x (wblocks*8)
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
    void fill() {
        while (nbits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) { p += 2; }
                    else { hit_marker = true; b = 0; }          // a marker: feed zeros, leave it in place
                } else {
                    ++p;
                }
            }
            acc |= (uint32_t)b << (24 - nbits);
            nbits += 8;
        }
    }
    int get(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    int bit() { return get(1); }
};

inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

int decode_symbol(BitReader& br, const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    fail("JPEG: bad Huffman code");
}

struct Jpeg {
    const uint8_t* d;
    size_t n;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false;
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    Component comp[4];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int restart_interval = 0;
    bool adobe = false;
    int adobe_transform = 0;
    int orientation = 1;

    static int u16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

    void parse_exif(const uint8_t* p, int len) {
        if (len < 14 || memcmp(p, "Exif\0\0", 6) != 0) return;
        const uint8_t* t = p + 6;
        const int tl = len - 6;
        const bool le = t[0] == 'I' && t[1] == 'I';
        if (!le && !(t[0] == 'M' && t[1] == 'M')) return;
        auto r16 = [&](int o) { return le ? t[o] | (t[o + 1] << 8) : (t[o] << 8) | t[o + 1]; };
        auto r32 = [&](int o) {
            return le ? (uint32_t)t[o] | ((uint32_t)t[o + 1] << 8) | ((uint32_t)t[o + 2] << 16) | ((uint32_t)t[o + 3] << 24)
                      : ((uint32_t)t[o] << 24) | ((uint32_t)t[o + 1] << 16) | ((uint32_t)t[o + 2] << 8) | (uint32_t)t[o + 3];
        };
        if (r16(2) != 42) return;
        const uint32_t ifd = r32(4);
        if (ifd + 2 > (uint32_t)tl) return;
        const int cnt = r16((int)ifd);
        for (int i = 0; i < cnt; ++i) {
            const int e = (int)ifd + 2 + 12 * i;
            if (e + 12 > tl) return;
            if (r16(e) == 0x0112) {
                const int v = r16(e + 8);
                if (v >= 1 && v <= 8) orientation = v;
                return;
            }
        }
    }

    void read_sof(const uint8_t* p, int len) {
        if (len < 6) fail("JPEG: short SOF");
        if (p[0] != 8) fail("JPEG: only 8-bit precision is supported");
        height = u16(p + 1); width = u16(p + 3); ncomp = p[5];
        if (width <= 0 || height <= 0) fail("JPEG: empty image");
        check_size(width, height, "JPEG");
        if (ncomp != 1 && ncomp != 3) fail("JPEG: only 1- and 3-component images are supported");
        if (len < 6 + 3 * ncomp) fail("JPEG: short SOF");
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i] & 3;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) fail("JPEG: bad sampling factor");
            hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
        }
        mcux = (width + 8 * hmax - 1) / (8 * hmax);
        mcuy = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.wblocks = mcux * c.h; c.hblocks = mcuy * c.v;
            c.dw = (int)(((long)width * c.h + hmax - 1) / hmax);
            c.dh = (int)(((long)height * c.v + vmax - 1) / vmax);
            c.coef.assign((size_t)c.wblocks * c.hblocks * 64, 0);
        }
    }

    // ---- one scan ------------------------------------------------------------------------------
    int eobrun = 0;

    void decode_block_baseline(BitReader& br, int16_t* blk, const Huff& hd, const Huff& ha, int& pred) {
        const int t = decode_symbol(br, hd);
        if (t > 15) fail("JPEG: bad DC category");              // (a damaged DHT can hold any byte as a symbol; 8-bit files use 0..11)
        const int diff = t ? extend(br.get(t), t) : 0;
        pred += diff;
        blk[0] = (int16_t)pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(br, ha);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) fail("JPEG: coefficient index out of range");
                blk[kZigzag[k]] = (int16_t)extend(br.get(s), s);
                ++k;
            }
        }
    }
    void decode_dc_first(BitReader& br, int16_t* blk, const Huff& hd, int& pred, int al) {
        const int t = decode_symbol(br, hd);
        if (t > 15) fail("JPEG: bad DC category");
        const int diff = t ? extend(br.get(t), t) : 0;
        pred += diff;
        blk[0] = (int16_t)(pred * (1 << al));
    }
    void decode_dc_refine(BitReader& br, int16_t* blk, int al) {
        if (br.bit()) blk[0] |= (int16_t)(1 << al);
    }
    void decode_ac_first(BitReader& br, int16_t* blk, const Huff& ha, int ss, int se, int al) {
        if (eobrun > 0) { --eobrun; return; }
        for (int k = ss; k <= se;) {
            const int rs = decode_symbol(br, ha);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eobrun = (1 << r) - 1;
                    if (r) eobrun += br.get(r);
                    break;
                }
                k += 16;
            } else {
                k += r;
                if (k > 63) fail("JPEG: coefficient index out of range");
                blk[kZigzag[k]] = (int16_t)(extend(br.get(s), s) * (1 << al));
                ++k;
            }
        }
    }
    void decode_ac_refine(BitReader& br, int16_t* blk, const Huff& ha, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se;) {
                const int rs = decode_symbol(br, ha);
                int r = rs >> 4;
                const int s = rs & 15;
                int val = 0;
                if (s == 0) {
                    if (r < 15) {
                        eobrun = 1 << r;
                        if (r) eobrun += br.get(r);
                        break;
                    }
                } else {
                    if (s != 1) fail("JPEG: bad refinement symbol");
                    val = br.bit() ? p1 : m1;
                }
                // skip r zero-history coefficients, correcting the non-zero ones on the way
                for (; k <= se; ++k) {
                    int16_t& c = blk[kZigzag[k]];
                    if (c != 0) {
                        if (br.bit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
                    } else {
                        if (r == 0) break;
                        --r;
                    }
                }
                if (val && k <= se) blk[kZigzag[k]] = (int16_t)val;
                ++k;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t& c = blk[kZigzag[k]];
                if (c != 0 && br.bit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
            }
            --eobrun;
        }
    }

    // returns the position after the scan's entropy-coded data
    size_t read_scan(size_t pos, int len) {
        const uint8_t* p = d + pos;
        if (len < 1) fail("JPEG: short SOS");
        const int ns = p[0];
        if (ns < 1 || ns > ncomp || len < 1 + 2 * ns + 3) fail("JPEG: bad SOS");
        int ci[4], td[4], ta[4];
        for (int i = 0; i < ns; ++i) {
            const int id = p[1 + 2 * i];
            ci[i] = -1;
            for (int c = 0; c < ncomp; ++c)
                if (comp[c].id == id) ci[i] = c;
            if (ci[i] < 0) fail("JPEG: scan names an unknown component");
            td[i] = p[2 + 2 * i] >> 4; ta[i] = p[2 + 2 * i] & 15;
            if (td[i] > 3 || ta[i] > 3) fail("JPEG: bad table selector");
            Component& c = comp[ci[i]];
            if (!c.quant_latched) {                               // libjpeg latches the table at a component's first scan
                if (!qt_present[c.tq]) fail("JPEG: missing quantisation table");
                memcpy(c.quant, qt[c.tq], sizeof(c.quant));
                c.quant_latched = true;
            }
        }
        const int ss = p[1 + 2 * ns], se = p[2 + 2 * ns], ah = p[3 + 2 * ns] >> 4, al = p[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) fail("JPEG: bad progressive scan");
        }
        for (int i = 0; i < ns; ++i) {
            const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
            if (need_dc && !(progressive && ah > 0) && !dc[td[i]].present) fail("JPEG: missing DC Huffman table");
            if (need_ac && !ac[ta[i]].present) fail("JPEG: missing AC Huffman table");
        }

        BitReader br{d + pos + len, d + n};
        int pred[4] = {0, 0, 0, 0};
        eobrun = 0;
        int restarts_left = restart_interval, next_rst = 0;
        auto handle_restart = [&]() {
            if (!restart_interval) return;
            if (restarts_left == 0) {
                // expect RSTn at br.p (byte aligned)
                br.reset();
                const uint8_t* q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                    if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) break;      // some other marker: give up resync
                    ++q;
                }
                if (q + 1 < br.end && q[0] == 0xFF && q[1] == 0xD0 + next_rst) q += 2;
                br.p = q;
                next_rst = (next_rst + 1) & 7;
                restarts_left = restart_interval;
                pred[0] = pred[1] = pred[2] = pred[3] = 0;
                eobrun = 0;
            }
            --restarts_left;
        };

        auto do_block = [&](int i, int bx, int by) {
            Component& c = comp[ci[i]];
            int16_t* blk = &c.coef[((size_t)by * c.wblocks + bx) * 64];
            if (!progressive) decode_block_baseline(br, blk, dc[td[i]], ac[ta[i]], pred[i]);
            else if (ss == 0) { if (ah == 0) decode_dc_first(br, blk, dc[td[i]], pred[i], al); else decode_dc_refine(br, blk, al); }
            else { if (ah == 0) decode_ac_first(br, blk, ac[ta[i]], ss, se, al); else decode_ac_refine(br, blk, ac[ta[i]], ss, se, al); }
        };

        if (ns == 1) {                                            // non-interleaved: the component's own block raster
            Component& c = comp[ci[0]];
            const int bw = (c.dw + 7) / 8, bh = (c.dh + 7) / 8;
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) {
                    handle_restart();
                    do_block(0, bx, by);
                }
        } else {
            for (int my = 0; my < mcuy; ++my)
                for (int mx = 0; mx < mcux; ++mx) {
                    handle_restart();
                    for (int i = 0; i < ns; ++i) {
                        const Component& c = comp[ci[i]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) do_block(i, mx * c.h + x, my * c.v + y);
                    }
                }
        }
        // continue after the entropy-coded segment: scan forward to the next real marker
        const uint8_t* q = br.p;
        while (q + 1 < d + n) {
            if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7)) break;
            ++q;
        }
        return (size_t)(q - d);
    }

    // ---- inverse DCT: jidctint.c "islow" (13-bit constants, 2 extra bits between the passes) ----
    static inline uint8_t range_limit(int x) {                   // libjpeg's centred range-limit table, index & 1023
        const int i = x & 1023;
        if (i < 128) return (uint8_t)(i + 128);
        if (i < 512) return 255;
        if (i < 896) return 0;
        return (uint8_t)(i - 896);
    }
    static void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
        constexpr int CB = 13, P1 = 2;
        constexpr long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
                       F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        long ws[64];
        auto descale = [](long x, int nb) { return (x + (1L << (nb - 1))) >> nb; };
        for (int c = 0; c < 8; ++c) {
            auto v = [&](int r) { return (long)in[r * 8 + c] * q[r * 8 + c]; };
            long z2 = v(2), z3 = v(6);
            long z1 = (z2 + z3) * F0_541;
            long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            z2 = v(0); z3 = v(4);
            long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = v(7); tmp1 = v(5); tmp2 = v(3); tmp3 = v(1);
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            ws[0 * 8 + c] = descale(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = descale(tmp10 - tmp3, CB - P1);
            ws[1 * 8 + c] = descale(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = descale(tmp11 - tmp2, CB - P1);
            ws[2 * 8 + c] = descale(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = descale(tmp12 - tmp1, CB - P1);
            ws[3 * 8 + c] = descale(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; ++r) {
            const long* w = ws + r * 8;
            uint8_t* o = out + (size_t)r * stride;
            long z2 = w[2], z3 = w[6];
            long z1 = (z2 + z3) * F0_541;
            long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
            long tmp0 = (w[0] + w[4]) * (1L << CB), tmp1 = (w[0] - w[4]) * (1L << CB);
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1_175;
            tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
            z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            constexpr int SH = CB + P1 + 3;
            o[0] = range_limit((int)descale(tmp10 + tmp3, SH)); o[7] = range_limit((int)descale(tmp10 - tmp3, SH));
            o[1] = range_limit((int)descale(tmp11 + tmp2, SH)); o[6] = range_limit((int)descale(tmp11 - tmp2, SH));
            o[2] = range_limit((int)descale(tmp12 + tmp1, SH)); o[5] = range_limit((int)descale(tmp12 - tmp1, SH));
            o[3] = range_limit((int)descale(tmp13 + tmp0, SH)); o[4] = range_limit((int)descale(tmp13 - tmp0, SH));
        }
    }

    // ---- chroma up-sampling to full resolution: jdsample.c --------------------------------------
    // src: c.plane (stride sw) with real size dw x dh; result: width x height samples
    std::vector<uint8_t> upsample(const Component& c) const {
        const int sw = c.wblocks * 8;
        const uint8_t* src = c.plane.data();
        std::vector<uint8_t> out((size_t)width * height);
        const int hx = hmax / c.h, vx = vmax / c.v;
        if (hmax % c.h || vmax % c.v) fail("JPEG: fractional sampling ratio");
        const int dw = c.dw, dh = c.dh;
        auto row = [&](int y) { return src + (size_t)std::min(std::max(y, 0), dh - 1) * sw; };    // edge rows replicate
        if (hx == 1 && vx == 1) {
            for (int y = 0; y < height; ++y) memcpy(&out[(size_t)y * width], row(y), width);
        } else if (hx == 2 && vx == 1 && dw > 2) {                // h2v1_fancy_upsample
            std::vector<uint8_t> line((size_t)2 * dw);
            for (int y = 0; y < height; ++y) {
                const uint8_t* in = row(y);
                line[0] = in[0];
                line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int i = 1; i < dw - 1; ++i) {
                    const int v = in[i] * 3;
                    line[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
                    line[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
                }
                line[2 * dw - 2] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
                line[2 * dw - 1] = in[dw - 1];
                memcpy(&out[(size_t)y * width], line.data(), width);
            }
        } else if (hx == 2 && vx == 2 && dw > 2) {                // h2v2_fancy_upsample
            std::vector<uint8_t> line((size_t)2 * dw);
            for (int y = 0; y < height; ++y) {
                const int sy = y >> 1;
                const uint8_t* in0 = row(sy);
                const uint8_t* in1 = row((y & 1) ? sy + 1 : sy - 1);
                int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                line[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
                line[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol; thiscol = nextcol;
                for (int i = 1; i < dw - 1; ++i) {
                    nextcol = in0[i + 1] * 3 + in1[i + 1];
                    line[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                    line[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol; thiscol = nextcol;
                }
                line[2 * dw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                line[2 * dw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
                memcpy(&out[(size_t)y * width], line.data(), width);
            }
        } else if (hx == 1 && vx == 2) {                          // h1v2_fancy_upsample (libjpeg-turbo)
            for (int y = 0; y < height; ++y) {
                const int sy = y >> 1;
                const uint8_t* in0 = row(sy);
                const uint8_t* in1 = row((y & 1) ? sy + 1 : sy - 1);
                const int bias = (y & 1) ? 2 : 1;
                for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            }
        } else {                                                  // int_upsample / h2v1 / h2v2 box replication
            for (int y = 0; y < height; ++y) {
                const uint8_t* in = row(y / vx);
                for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = in[x / hx];
            }
        }
        return out;
    }

    Image decode() {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) fail("JPEG: no SOI");
        size_t pos = 2;
        bool have_sof = false, done = false;
        while (!done) {
            while (pos < n && d[pos] != 0xFF) ++pos;
            while (pos < n && d[pos] == 0xFF) ++pos;
            if (pos >= n) break;
            const int m = d[pos++];
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (pos + 2 > n) fail("JPEG: truncated marker");
            const int len = u16(d + pos) - 2;
            const uint8_t* p = d + pos + 2;
            if (len < 0 || pos + 2 + (size_t)len > n) fail("JPEG: truncated segment");
            switch (m) {
                case 0xC0: case 0xC1: case 0xC2:
                    if (have_sof) fail("JPEG: more than one frame");
                    progressive = m == 0xC2;
                    read_sof(p, len);
                    have_sof = true;
                    break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    fail("JPEG: lossless / hierarchical / arithmetic-coded files are not supported");
                case 0xC4: {
                    int o = 0;
                    while (o + 17 <= len) {
                        const int tc = p[o] >> 4, th = p[o] & 15;
                        if (tc > 1 || th > 3) fail("JPEG: bad DHT");
                        Huff& h = tc ? ac[th] : dc[th];
                        int total = 0;
                        h.bits[0] = 0;
                        for (int i = 1; i <= 16; ++i) { h.bits[i] = p[o + i]; total += h.bits[i]; }
                        if (total > 256 || o + 17 + total > len) fail("JPEG: bad DHT");
                        memset(h.vals, 0, sizeof(h.vals));
                        memcpy(h.vals, p + o + 17, total);
                        h.build();
                        o += 17 + total;
                    }
                    break;
                }
                case 0xDB: {
                    int o = 0;
                    while (o < len) {
                        const int pq = p[o] >> 4, tq = p[o] & 15;
                        if (tq > 3 || pq > 1 || o + 1 + 64 * (pq + 1) > len) fail("JPEG: bad DQT");
                        for (int i = 0; i < 64; ++i) qt[tq][kZigzag[i]] = pq ? (uint16_t)u16(p + o + 1 + 2 * i) : p[o + 1 + i];
                        qt_present[tq] = true;
                        o += 1 + 64 * (pq + 1);
                    }
                    break;
                }
                case 0xDD:
                    if (len < 2) fail("JPEG: bad DRI");
                    restart_interval = u16(p);
                    break;
                case 0xE1: parse_exif(p, len); break;
                case 0xEE:
                    if (len >= 12 && memcmp(p, "Adobe", 5) == 0) { adobe = true; adobe_transform = p[11]; }
                    break;
                case 0xDA:
                    if (!have_sof) fail("JPEG: scan before frame header");
                    pos = read_scan(pos + 2, len);
                    continue;
                default: break;
            }
            pos += 2 + (size_t)len;
        }
        if (!have_sof) fail("JPEG: no frame header");

        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            if (!c.quant_latched) fail("JPEG: component without a scan");
            c.plane.assign((size_t)c.wblocks * 8 * c.hblocks * 8, 0);
            const int stride = c.wblocks * 8;
            for (int by = 0; by < c.hblocks; ++by)
                for (int bx = 0; bx < c.wblocks; ++bx)
                    idct_islow(&c.coef[((size_t)by * c.wblocks + bx) * 64], c.quant, &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
        }
        Image img;
        img.rows = height; img.cols = width;
        img.bgr.resize((size_t)width * height * 3);
        if (ncomp == 1) {
            const std::vector<uint8_t> y = upsample(comp[0]);
            for (size_t i = 0; i < y.size(); ++i) img.bgr[3 * i] = img.bgr[3 * i + 1] = img.bgr[3 * i + 2] = y[i];
        } else {
            const std::vector<uint8_t> p0 = upsample(comp[0]), p1 = upsample(comp[1]), p2 = upsample(comp[2]);
            const bool rgb = adobe ? adobe_transform == 0 : (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
            if (rgb) {
                for (size_t i = 0; i < p0.size(); ++i) { img.bgr[3 * i] = p2[i]; img.bgr[3 * i + 1] = p1[i]; img.bgr[3 * i + 2] = p0[i]; }
            } else {                                              // jdcolor.c ycc_rgb_convert, SCALEBITS = 16
                int cr_r[256], cb_b[256];
                long cr_g[256], cb_g[256];
                auto fix = [](double x) { return (long)(x * 65536.0 + 0.5); };
                for (int i = 0; i < 256; ++i) {
                    const long x = i - 128;
                    cr_r[i] = (int)((fix(1.40200) * x + 32768) >> 16);
                    cb_b[i] = (int)((fix(1.77200) * x + 32768) >> 16);
                    cr_g[i] = -fix(0.71414) * x;
                    cb_g[i] = -fix(0.34414) * x + 32768;
                }
                auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
                for (size_t i = 0; i < p0.size(); ++i) {
                    const int y = p0[i], cb = p1[i], cr = p2[i];
                    img.bgr[3 * i + 2] = clamp(y + cr_r[cr]);
                    img.bgr[3 * i + 1] = clamp(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
                    img.bgr[3 * i] = clamp(y + cb_b[cb]);
                }
            }
        }
        return img;
    }
};

Image apply_orientation(Image in, int o) {
    if (o <= 1 || o > 8) return in;
    const int R = in.rows, C = in.cols;
    Image out;
    const bool swap = o >= 5;
    out.rows = swap ? C : R; out.cols = swap ? R : C;
    out.bgr.resize(in.bgr.size());
    for (int y = 0; y < out.rows; ++y)
        for (int x = 0; x < out.cols; ++x) {
            int sy, sx;
            switch (o) {
                case 2: sy = y; sx = C - 1 - x; break;               // mirror horizontal
                case 3: sy = R - 1 - y; sx = C - 1 - x; break;       // rotate 180
                case 4: sy = R - 1 - y; sx = x; break;               // mirror vertical
                case 5: sy = x; sx = y; break;                       // transpose
                case 6: sy = R - 1 - x; sx = y; break;               // rotate 90 clockwise
                case 7: sy = R - 1 - x; sx = C - 1 - y; break;       // transverse
                default: sy = x; sx = C - 1 - y; break;              // 8: rotate 270 clockwise
            }
            memcpy(&out.bgr[((size_t)y * out.cols + x) * 3], &in.bgr[((size_t)sy * C + sx) * 3], 3);
        }
    return out;
}

// =====================================================================================================
// PNG
// =====================================================================================================
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void png_unfilter(uint8_t* cur, const uint8_t* prev, int ft, size_t rowbytes, int bpp) {
    switch (ft) {
        case 0: break;
        case 1:
            for (size_t i = bpp; i < rowbytes; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            if (prev) for (size_t i = 0; i < rowbytes; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
            break;
        case 3:
            for (size_t i = 0; i < rowbytes; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < rowbytes; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                cur[i] = (uint8_t)(cur[i] + (pa <= pb && pa <= pc ? a : pb <= pc ? b : c));
            }
            break;
        default: fail("PNG: bad filter type");
    }
}

Image decode_png(const uint8_t* d, size_t n) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(d, sig, 8) != 0) fail("PNG: bad signature");
    size_t pos = 8;
    int W = 0, H = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_hdr = false;
    while (pos + 12 <= n) {
        const uint32_t len = be32(d + pos);
        const uint8_t* type = d + pos + 4;
        const uint8_t* body = d + pos + 8;
        if (pos + 12 + (size_t)len > n) fail("PNG: truncated chunk");
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) fail("PNG: bad IHDR");
            W = (int)be32(body); H = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (W <= 0 || H <= 0 || body[10] != 0 || body[11] != 0 || interlace > 1) fail("PNG: unsupported header");
            check_size(W, H, "PNG");
            have_hdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_hdr) fail("PNG: no IHDR");
    int ch;
    switch (ctype) {
        case 0: ch = 1; break;
        case 2: ch = 3; break;
        case 3: ch = 1; break;
        case 4: ch = 2; break;
        case 6: ch = 4; break;
        default: fail("PNG: bad colour type");
    }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) fail("PNG: bad bit depth");
    if (ctype == 3 && plte.size() < 3) fail("PNG: palette image without PLTE");
    const int bits_pp = ch * depth, bpp = std::max(1, bits_pp / 8);
    auto rowbytes = [&](int w) { return ((size_t)w * bits_pp + 7) / 8; };

    // pass geometry (Adam7 or the single full pass)
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int npass = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int p = 0; p < npass; ++p) {
        const int pw = interlace ? (W - xs[p] + dx[p] - 1) / dx[p] : W, ph = interlace ? (H - ys[p] + dy[p] - 1) / dy[p] : H;
        if (pw > 0 && ph > 0) raw_size += (rowbytes(pw) + 1) * (size_t)ph;
    }
    std::vector<uint8_t> raw(raw_size);
    {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) fail("PNG: zlib init failed");
        zs.next_in = idat.data(); zs.avail_in = (uInt)idat.size();
        zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = raw.size() - zs.avail_out;
        inflateEnd(&zs);
        if ((rc != Z_STREAM_END && rc != Z_OK && rc != Z_BUF_ERROR) || got != raw.size()) fail("PNG: corrupt image data");
    }

    Image img;
    img.rows = H; img.cols = W;
    img.bgr.assign((size_t)W * H * 3, 0);
    auto put = [&](int x, int y, const uint8_t* row, int i) {       // pixel i of an unfiltered row -> BGR
        uint8_t r, g, b;
        auto sample = [&](int s) -> int {                           // s-th sample of pixel i, reduced to 8 bits
            if (depth == 8) return row[(size_t)i * ch + s];
            if (depth == 16) return row[((size_t)i * ch + s) * 2];  // high byte (png_set_strip_16)
            const int per = 8 / depth, v = (row[i / per] >> (8 - depth * (i % per + 1))) & ((1 << depth) - 1);
            return v;
        };
        if (ctype == 3) {
            const int idx = sample(0);
            if ((size_t)idx * 3 + 2 < plte.size()) { r = plte[idx * 3]; g = plte[idx * 3 + 1]; b = plte[idx * 3 + 2]; }
            else { r = g = b = 0; }
        } else if (ctype == 0 || ctype == 4) {
            int v = sample(0);
            if (depth < 8) v = v * 255 / ((1 << depth) - 1);        // png_set_expand_gray_1_2_4_to_8
            r = g = b = (uint8_t)v;
        } else {
            r = (uint8_t)sample(0); g = (uint8_t)sample(1); b = (uint8_t)sample(2);
        }
        uint8_t* o = &img.bgr[((size_t)y * W + x) * 3];
        o[0] = b; o[1] = g; o[2] = r;
    };
    size_t off = 0;
    for (int p = 0; p < npass; ++p) {
        const int pw = interlace ? (W - xs[p] + dx[p] - 1) / dx[p] : W, ph = interlace ? (H - ys[p] + dy[p] - 1) / dy[p] : H;
        if (pw <= 0 || ph <= 0) continue;
        const size_t rb = rowbytes(pw);
        const uint8_t* prev = nullptr;
        for (int y = 0; y < ph; ++y) {
            uint8_t* line = &raw[off];
            png_unfilter(line + 1, prev, line[0], rb, bpp);
            prev = line + 1;
            for (int x = 0; x < pw; ++x) put(interlace ? xs[p] + x * dx[p] : x, interlace ? ys[p] + y * dy[p] : y, line + 1, x);
            off += rb + 1;
        }
    }
    return img;
}

// =====================================================================================================
// BMP, PPM / PGM
// =====================================================================================================
Image decode_bmp(const uint8_t* d, size_t n) {
    if (n < 54) fail("BMP: truncated header");
    auto le32 = [&](size_t o) { return (int32_t)((uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24)); };
    auto le16 = [&](size_t o) { return d[o] | (d[o + 1] << 8); };
    const size_t data_off = (uint32_t)le32(10);
    const int W = le32(18), Hs = le32(22), bpp = le16(28), comp = le32(30);
    if (le16(26) != 1 || (bpp != 24 && bpp != 32) || (comp != 0 && !(comp == 3 && bpp == 32)) || W <= 0 || Hs == 0) fail("BMP: only uncompressed 24/32-bit files are supported");
    const int H = abs(Hs);
    check_size(W, H, "BMP");
    const size_t stride = ((size_t)W * (bpp / 8) + 3) & ~(size_t)3;
    if (data_off + stride * H > n) fail("BMP: truncated pixel data");
    Image img;
    img.rows = H; img.cols = W;
    img.bgr.resize((size_t)W * H * 3);
    for (int y = 0; y < H; ++y) {
        const uint8_t* src = d + data_off + stride * (size_t)(Hs > 0 ? H - 1 - y : y);
        for (int x = 0; x < W; ++x) memcpy(&img.bgr[((size_t)y * W + x) * 3], src + (size_t)x * (bpp / 8), 3);
    }
    return img;
}

Image decode_pnm(const uint8_t* d, size_t n) {
    size_t pos = 2;
    auto next_int = [&]() {
        for (;;) {
            while (pos < n && isspace(d[pos])) ++pos;
            if (pos < n && d[pos] == '#') { while (pos < n && d[pos] != '\n') ++pos; continue; }
            break;
        }
        if (pos >= n || !isdigit(d[pos])) fail("PNM: bad header");
        long v = 0;
        while (pos < n && isdigit(d[pos])) { v = v * 10 + (d[pos++] - '0'); if (v > (1 << 30)) fail("PNM: bad header"); }
        return (int)v;
    };
    const int ch = d[1] == '6' ? 3 : 1;
    const int W = next_int(), H = next_int(), maxv = next_int();
    if (W <= 0 || H <= 0 || maxv <= 0 || maxv > 255) fail("PNM: only 8-bit binary files are supported");
    check_size(W, H, "PNM");
    ++pos;                                                        // the single whitespace byte after maxval
    if (pos + (size_t)W * H * ch > n) fail("PNM: truncated pixel data");
    Image img;
    img.rows = H; img.cols = W;
    img.bgr.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const uint8_t* s = d + pos + i * ch;
        if (ch == 3) { img.bgr[3 * i] = s[2]; img.bgr[3 * i + 1] = s[1]; img.bgr[3 * i + 2] = s[0]; }
        else { img.bgr[3 * i] = img.bgr[3 * i + 1] = img.bgr[3 * i + 2] = s[0]; }
    }
    return img;
}

Image decode_any(const uint8_t* d, size_t n) {
    if (n >= 3 && d[0] == 0xFF && d[1] == 0xD8) {
        Jpeg j;
        j.d = d; j.n = n;
        Image img = j.decode();
        return apply_orientation(std::move(img), j.orientation);
    }
    if (n >= 8 && d[0] == 0x89 && d[1] == 'P') return decode_png(d, n);
    if (n >= 2 && d[0] == 'B' && d[1] == 'M') return decode_bmp(d, n);
    if (n >= 2 && d[0] == 'P' && (d[1] == '6' || d[1] == '5')) return decode_pnm(d, n);
    fail("unrecognised image format (JPEG, PNG, BMP, PPM/PGM are supported)");
}

}  // namespace

extern "C" {

int fh_image_decode(const unsigned char* bytes, size_t n, unsigned char** bgr, int* rows, int* cols) {
    if (!bytes || !bgr || !rows || !cols) { fh::set_error("fh_image_decode: null argument"); return -1; }
    *bgr = nullptr; *rows = *cols = 0;
    try {
        Image img = decode_any(bytes, n);
        unsigned char* out = static_cast<unsigned char*>(malloc(img.bgr.size() ? img.bgr.size() : 1));
        if (!out) { fh::set_error("fh_image_decode: out of memory"); return -1; }
        memcpy(out, img.bgr.data(), img.bgr.size());
        *bgr = out; *rows = img.rows; *cols = img.cols;
        return 0;
    } catch (const DecodeError& e) {
        fh::set_error(e.msg);
    } catch (const std::exception& e) {
        fh::set_error(std::string("fh_image_decode: ") + e.what());
    }
    return -1;
}

int fh_imread(const char* path, unsigned char** bgr, int* rows, int* cols) {
    if (!path || !bgr || !rows || !cols) { fh::set_error("fh_imread: null argument"); return -1; }
    *bgr = nullptr; *rows = *cols = 0;
    FILE* f = fopen(path, "rb");
    if (!f) { fh::set_error(std::string("fh_imread: cannot open ") + path); return -1; }
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    return fh_image_decode(buf.data(), buf.size(), bgr, rows, cols);
}

void fh_image_free(unsigned char* bgr) { free(bgr); }

}  // extern "C"
