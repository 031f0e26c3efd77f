// face_detector.h — drop-in for the reference's src/face_detector.h: same struct FaceBox, same
// class name, method names, argument order, defaults and return types (reference
// src/face_detector.h:8-20).  The ONNX Runtime members are replaced by one opaque handle of the
// C ABI (include/facehip.h); copying is disabled (the reference's implicit copy double-frees
// its raw Ort::Session*), moving is allowed.
#pragma once
#include <string>
#include <vector>

#include "cv_compat.h"

struct fh_det;

struct FaceBox {
    cv::Rect box;
    float score;
    cv::Point2f landmarks[5];   // left eye, right eye, nose, left mouth corner, right mouth corner
};

class FaceDetector {
  public:
    FaceDetector();
    ~FaceDetector();
    FaceDetector(const FaceDetector&) = delete;
    FaceDetector& operator=(const FaceDetector&) = delete;
    FaceDetector(FaceDetector&& o) noexcept;
    FaceDetector& operator=(FaceDetector&& o) noexcept;

    bool loadModel(const std::string& modelPath);
    std::vector<FaceBox> detect(const cv::Mat& image, float scoreThreshold = 0.5f, float nmsThreshold = 0.4f);

    fh_det* handle() const { return h_; }       // for the batch / device entry points of facehip.h

  private:
    fh_det* h_;
};
