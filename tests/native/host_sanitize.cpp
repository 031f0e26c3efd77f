// AddressSanitizer / UndefinedBehaviorSanitizer harness for the product's HOST-side parsers (the only place where the library reads
// bytes it did not write): csrc/image_io.cpp (JPEG / PNG / BMP / PNM decoders = cv::imread's stand-in, reference src/main.cpp:42,71-72)
// and csrc/onnx_reader.cpp + csrc/plan.cpp (loadModel, reference src/face_detector.cpp:20-90).  GPU sanitizers are not available on the
// pool, so this runs on the CPU build:   g++ -fsanitize=address,undefined ... (tests/test_host_sanitize.py builds and runs it).
//
//   host_sanitize <golden dir>
// Every golden image and model is decoded as it is (must succeed), then as 200+ damaged variants each — truncations at many lengths,
// single-byte flips, 16-byte splats of 0x00 / 0xFF at seeded positions — which must come back as an error or a picture, never as a
// sanitizer report, a crash or an allocation of absurd size.
#include <dirent.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/facehip.h"
#include "../../facerecognizeonnx_amd/csrc/plan.h"

namespace fh {
static std::string g_last;
void set_error(const std::string& msg) { g_last = msg; }
}  // namespace fh

static std::vector<unsigned char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static std::vector<std::string> list(const std::string& dir, const char* suffix) {
    std::vector<std::string> out;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n.size() > strlen(suffix) && n.compare(n.size() - strlen(suffix), strlen(suffix), suffix) == 0) out.push_back(dir + "/" + n);
        }
        closedir(d);
    }
    return out;
}
static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

static int decode(const std::vector<unsigned char>& b, bool must_succeed, const std::string& what) {
    unsigned char* px = nullptr; int rows = 0, cols = 0;
    const int rc = fh_image_decode(b.data(), b.size(), &px, &rows, &cols);
    if (rc == 0) {
        if (!px || rows <= 0 || cols <= 0 || (long long)rows * cols > (1ll << 28)) { fprintf(stderr, "%s: bogus success %dx%d\n", what.c_str(), rows, cols); return 1; }
        volatile unsigned sum = 0;                                         // touch every byte the decoder says it owns
        for (size_t i = 0; i < (size_t)rows * cols * 3; i += 97) sum += px[i];
        (void)sum;
        fh_image_free(px);
    } else if (must_succeed) {
        fprintf(stderr, "%s: decode failed: %s\n", what.c_str(), fh::g_last.c_str());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: host_sanitize <tests/golden>\n"); return 2; }
    const std::string golden = argv[1];
    int bad = 0, images = 0, variants = 0, models = 0;
    for (const char* ext : {".jpg", ".jpeg", ".png", ".bmp", ".ppm", ".pgm"})
        for (const std::string& p : list(golden + "/images", ext)) {
            const std::vector<unsigned char> b = slurp(p);
            ++images;
            bad += decode(b, true, p);
            for (int t = 0; t < 48; ++t) {                                 // truncations: dense near the header, sparse over the body
                const size_t n = t < 24 ? (size_t)t * 7 : (size_t)((double)b.size() * (t - 23) / 25.0);
                if (n >= b.size()) continue;
                bad += decode(std::vector<unsigned char>(b.begin(), b.begin() + n), false, p + " truncated"); ++variants;
            }
            for (int t = 0; t < 120; ++t) {                                // byte flips, biased towards the first kilobyte (headers, tables)
                std::vector<unsigned char> c = b;
                const size_t pos = (t & 1) ? rnd() % std::min<size_t>(c.size(), 1024) : rnd() % c.size();
                c[pos] ^= (unsigned char)(1u << (rnd() % 8));
                if (t % 3 == 0) c[pos] = (unsigned char)rnd();
                bad += decode(c, false, p + " flipped"); ++variants;
            }
            for (int t = 0; t < 40; ++t) {                                 // splats
                std::vector<unsigned char> c = b;
                const size_t pos = rnd() % c.size();
                for (size_t i = pos; i < std::min(c.size(), pos + 16); ++i) c[i] = (t & 1) ? 0xFF : 0x00;
                bad += decode(c, false, p + " splat"); ++variants;
            }
        }
    for (const std::string& p : list(golden, ".onnx")) {
        ++models;
        const bool det = p.find("scrfd") != std::string::npos;
        try {
            const fh::OnnxModel m = fh::load_onnx(p);
            const fh::Plan pl = fh::build_plan(m, det ? 640 : 112, det ? 640 : 112);
            if (pl.ops.empty()) { fprintf(stderr, "%s: empty plan\n", p.c_str()); ++bad; }
        } catch (const std::exception& e) {
            fprintf(stderr, "%s: %s\n", p.c_str(), e.what()); ++bad;
        }
        const std::vector<unsigned char> b = slurp(p);
        const std::string tmp = std::string(argc > 2 ? argv[2] : "/tmp") + "/host_sanitize_variant.onnx";
        for (int t = 0; t < 60; ++t) {                                     // damaged model files: must throw (or load), not crash
            std::vector<unsigned char> c = b;
            if (t < 20) c.resize(t < 10 ? (size_t)t * 5 : (size_t)((double)b.size() * (t - 9) / 11.0));
            else {
                const size_t lim = t < 45 ? std::min<size_t>(c.size(), 4096) : c.size();   // node / graph headers sit at the front
                for (int k = 0; k < 1 + t % 3; ++k) c[rnd() % lim] = (unsigned char)rnd();
            }
            { std::ofstream o(tmp, std::ios::binary); o.write(reinterpret_cast<const char*>(c.data()), (std::streamsize)c.size()); }
            try {
                const fh::OnnxModel m = fh::load_onnx(tmp);
                (void)fh::build_plan(m, det ? 640 : 112, det ? 640 : 112);
            } catch (const std::exception&) {
            }
            ++variants;
        }
        remove(tmp.c_str());
    }
    printf("host_sanitize: %d images, %d models, %d damaged variants, %d failures\n", images, models, variants, bad);
    return bad ? 1 : (images >= 20 && models >= 4 ? 0 : 3);
}
