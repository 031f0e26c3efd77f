// shim_demo.cpp — the reference's `compare` flow (src/main.cpp:67-134: detect on two images,
// extractFeature of faces[0] of each, compareFaces, threshold 0.6) written against the drop-in
// headers: cv::imread -> detect -> extractFeature -> compareFaces, with text output instead of cv::imshow.
// usage: shim_demo det.onnx rec.onnx image1 image2 [score_thr]        (JPEG / PNG / BMP / PPM)
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "face_detector.h"
#include "face_recognizer.h"

int main(int argc, char** argv) {
    if (argc < 5) { std::cerr << "usage: shim_demo det.onnx rec.onnx image1 image2 [score_thr]\n"; return 2; }
    const float thr = argc > 5 ? (float)atof(argv[5]) : 0.5f;
    FaceDetector detector;
    if (!detector.loadModel(argv[1])) return -1;                       // main.cpp:274-278
    FaceRecognizer recognizer;
    if (!recognizer.loadModel(argv[2])) return -1;                     // main.cpp:280-284
    cv::Mat a = cv::imread(argv[3]), b = cv::imread(argv[4]);          // main.cpp:71-72
    if (a.empty() || b.empty()) { std::cerr << "Cannot read image" << std::endl; return -1; }   // main.cpp:74-77
    auto fa = detector.detect(a, thr), fb = detector.detect(b, thr);   // main.cpp:88-89
    printf("faces %zu %zu\n", fa.size(), fb.size());
    if (fa.empty() || fb.empty()) return 0;
    printf("box %d %d %d %d score %.6f\n", fa[0].box.x, fa[0].box.y, fa[0].box.width, fa[0].box.height, fa[0].score);
    auto f1 = recognizer.extractFeature(a, fa[0]), f2 = recognizer.extractFeature(b, fb[0]);   // main.cpp:101,104
    printf("dim %zu %zu\n", f1.size(), f2.size());
    const float sim = recognizer.compareFaces(f1, f2);                 // main.cpp:114
    printf("similarity %.6f %s\n", sim, sim > 0.6f ? "same" : "different");                    // main.cpp:118
    printf("self %.6f\n", recognizer.compareFaces(f1, f1));
    std::vector<float> emb = recognizer.extractFeatureSimple(a);       // main.cpp:155
    printf("simple %zu\n", emb.size());
    printf("f1");
    for (int i = 0; i < 8 && i < (int)f1.size(); ++i) printf(" %.6f", f1[i]);
    printf("\n");
    return 0;
}
