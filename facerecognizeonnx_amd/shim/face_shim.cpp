// face_shim.cpp — thin host shim: reference class API -> C ABI (include/facehip.h).
// Error mapping follows the reference: loadModel -> false + message on std::cerr
// (src/face_detector.cpp:86-89), detect / extractFeature -> empty vector + std::cerr
// (src/face_detector.cpp:142-167,217-219; src/face_recognizer.cpp:239-267,299-301),
// compareFaces -> 0.0f on size mismatch (src/face_recognizer.cpp:321-323).  Unlike the
// reference nothing is printed on the success path (detect() printed >= 6
// lines per call); set FACEHIP_VERBOSE=1 to get the load banner.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <utility>

#include "../../include/facehip.h"
#include "face_detector.h"
#include "face_recognizer.h"

namespace {
bool verbose() { const char* v = std::getenv("FACEHIP_VERBOSE"); return v && *v && *v != '0'; }
}  // namespace

FaceDetector::FaceDetector() : h_(nullptr) {}
FaceDetector::~FaceDetector() { if (h_) fh_det_destroy(h_); }
FaceDetector::FaceDetector(FaceDetector&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
FaceDetector& FaceDetector::operator=(FaceDetector&& o) noexcept {
    if (this != &o) { if (h_) fh_det_destroy(h_); h_ = o.h_; o.h_ = nullptr; }
    return *this;
}

bool FaceDetector::loadModel(const std::string& modelPath) {
    if (h_) { fh_det_destroy(h_); h_ = nullptr; }
    h_ = fh_det_create(modelPath.c_str());
    if (!h_) {
        std::cerr << "Error loading face detector model: " << fh_last_error() << std::endl;
        return false;
    }
    if (verbose()) {
        int w = 0, h = 0;
        fh_det_input_size(h_, &w, &h);
        std::cout << "Face detector model loaded successfully!\nUsing input size: " << w << "x" << h << std::endl;
    }
    return true;
}

std::vector<FaceBox> FaceDetector::detect(const cv::Mat& image, float scoreThreshold, float nmsThreshold) {
    std::vector<FaceBox> faces;
    if (!h_) { std::cerr << "Model not loaded!" << std::endl; return faces; }
    if (image.empty()) { std::cerr << "Input image is empty!" << std::endl; return faces; }
    // the reference returns every post-NMS box (std::vector, src/face_detector.cpp:376-383): size the buffer for the case that
    // every candidate row survives (16 800 anchors for det_500m) instead of truncating at a fixed count
    const int cap = fh_det_num_anchors(h_) > 0 ? fh_det_num_anchors(h_) : 16800;
    std::vector<fh_face> buf((size_t)cap);
    const int n = fh_det_detect(h_, image.data, image.rows, image.cols, (int)image.step, scoreThreshold, nmsThreshold, buf.data(), cap);
    if (n < 0) { std::cerr << "Error during inference: " << fh_last_error() << std::endl; return faces; }
    faces.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        faces[i].box = cv::Rect(buf[i].x, buf[i].y, buf[i].w, buf[i].h);
        faces[i].score = buf[i].score;
        for (int j = 0; j < 5; ++j) faces[i].landmarks[j] = cv::Point2f(buf[i].lm[2 * j], buf[i].lm[2 * j + 1]);
    }
    return faces;
}

FaceRecognizer::FaceRecognizer() : h_(nullptr) {}
FaceRecognizer::~FaceRecognizer() { if (h_) fh_rec_destroy(h_); }
FaceRecognizer::FaceRecognizer(FaceRecognizer&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
FaceRecognizer& FaceRecognizer::operator=(FaceRecognizer&& o) noexcept {
    if (this != &o) { if (h_) fh_rec_destroy(h_); h_ = o.h_; o.h_ = nullptr; }
    return *this;
}

bool FaceRecognizer::loadModel(const std::string& modelPath) {
    if (h_) { fh_rec_destroy(h_); h_ = nullptr; }
    h_ = fh_rec_create(modelPath.c_str());
    if (!h_) {
        std::cerr << "Error loading face recognizer model: " << fh_last_error() << std::endl;
        return false;
    }
    if (verbose()) std::cout << "Face recognizer model loaded successfully!" << std::endl;
    return true;
}

std::vector<float> FaceRecognizer::extractFeature(const cv::Mat& image, const FaceBox& face) {
    std::vector<float> feature;
    if (!h_) { std::cerr << "Model not loaded!" << std::endl; return feature; }
    if (image.empty()) { std::cerr << "Input image is empty!" << std::endl; return feature; }
    fh_face f;
    f.x = face.box.x; f.y = face.box.y; f.w = face.box.width; f.h = face.box.height; f.score = face.score;
    for (int j = 0; j < 5; ++j) { f.lm[2 * j] = face.landmarks[j].x; f.lm[2 * j + 1] = face.landmarks[j].y; }
    feature.resize((size_t)fh_rec_feature_dim(h_));
    const int n = fh_rec_extract(h_, image.data, image.rows, image.cols, (int)image.step, &f, feature.data(), (int)feature.size());
    if (n == 0) std::cerr << "Face alignment failed!" << std::endl;
    if (n < 0) std::cerr << "Error during feature extraction: " << fh_last_error() << std::endl;
    feature.resize(n > 0 ? (size_t)n : 0);
    return feature;
}

std::vector<float> FaceRecognizer::extractFeatureSimple(const cv::Mat& image) {
    std::vector<float> feature;
    if (!h_) { std::cerr << "Model not loaded!" << std::endl; return feature; }
    if (image.empty()) { std::cerr << "Input image is empty!" << std::endl; return feature; }
    feature.resize((size_t)fh_rec_feature_dim(h_));
    const int n = fh_rec_extract_simple(h_, image.data, image.rows, image.cols, (int)image.step, feature.data(), (int)feature.size());
    if (n < 0) std::cerr << "Error during feature extraction: " << fh_last_error() << std::endl;
    feature.resize(n > 0 ? (size_t)n : 0);
    return feature;
}

float FaceRecognizer::compareFaces(const std::vector<float>& feature1, const std::vector<float>& feature2) {
    return fh_compare(feature1.data(), (int)feature1.size(), feature2.data(), (int)feature2.size());
}
