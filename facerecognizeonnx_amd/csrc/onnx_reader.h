// onnx_reader.h — dependency-free ONNX (protobuf wire format) reader.
//
// Replaces what `new Ort::Session(env, path, opts)` parses for the reference
// (reference src/face_detector.cpp:20-90, src/face_recognizer.cpp:21-91).  Only the closed op
// set of the two face graphs is interpreted later (plan.cpp); the reader itself is generic.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace fh {

struct OnnxTensor {
    std::string name;
    std::vector<int64_t> dims;
    int dtype = 1;                 // 1 = float32, 7 = int64 (TensorProto.DataType)
    std::vector<float> f;          // float payload (float32 / converted double)
    std::vector<int64_t> i;        // integer payload (int64 / int32)
    size_t numel() const { size_t n = 1; for (auto d : dims) n *= (size_t)d; return n; }
};

struct OnnxAttr {
    int64_t i = 0;
    float f = 0.f;
    std::string s;
    std::vector<int64_t> ints;
    std::vector<float> floats;
    OnnxTensor t;
};

struct OnnxNode {
    std::string op, name;
    std::vector<std::string> inputs, outputs;
    std::map<std::string, OnnxAttr> attrs;
    int64_t attr_i(const std::string& k, int64_t dflt) const {
        auto it = attrs.find(k); return it == attrs.end() ? dflt : it->second.i;
    }
    float attr_f(const std::string& k, float dflt) const {
        auto it = attrs.find(k); return it == attrs.end() ? dflt : it->second.f;
    }
    std::vector<int64_t> attr_ints(const std::string& k) const {
        auto it = attrs.find(k); return it == attrs.end() ? std::vector<int64_t>{} : it->second.ints;
    }
    std::string attr_s(const std::string& k, const std::string& dflt) const {
        auto it = attrs.find(k); return it == attrs.end() ? dflt : it->second.s;
    }
};

struct OnnxValueInfo {
    std::string name;
    std::vector<int64_t> shape;    // -1 for dynamic (dim_param), as ORT reports it
};

struct OnnxModel {
    std::vector<OnnxNode> nodes;
    std::map<std::string, OnnxTensor> inits;
    std::vector<OnnxValueInfo> inputs;   // graph inputs that are not initializers
    std::vector<OnnxValueInfo> outputs;
};

// Throws std::runtime_error on malformed / unreadable files.
OnnxModel load_onnx(const std::string& path);

}  // namespace fh
