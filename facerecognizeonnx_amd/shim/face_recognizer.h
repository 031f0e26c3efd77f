// face_recognizer.h — drop-in for the reference's src/face_recognizer.h (same public surface:
// reference src/face_recognizer.h:11-17) over the C ABI of include/facehip.h.
#pragma once
#include <string>
#include <vector>

#include "cv_compat.h"
#include "face_detector.h"

struct fh_rec;

class FaceRecognizer {
  public:
    FaceRecognizer();
    ~FaceRecognizer();
    FaceRecognizer(const FaceRecognizer&) = delete;
    FaceRecognizer& operator=(const FaceRecognizer&) = delete;
    FaceRecognizer(FaceRecognizer&& o) noexcept;
    FaceRecognizer& operator=(FaceRecognizer&& o) noexcept;

    bool loadModel(const std::string& modelPath);
    std::vector<float> extractFeature(const cv::Mat& image, const FaceBox& face);
    std::vector<float> extractFeatureSimple(const cv::Mat& image);
    float compareFaces(const std::vector<float>& feature1, const std::vector<float>& feature2);

    fh_rec* handle() const { return h_; }

  private:
    fh_rec* h_;
};
