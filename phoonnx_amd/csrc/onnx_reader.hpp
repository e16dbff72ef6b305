// onnx_reader.hpp — dependency-free reader for the subset of ONNX protobuf the PyTorch
// exporter writes (phoonnx_train/export_onnx.py:318-327).  Host-side C++17.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace vitsmi {

struct OnnxTensor {
    std::string name;
    int dtype = 0;                 // onnx TensorProto.DataType (1 = float32, 7 = int64)
    std::vector<int64_t> dims;
    const uint8_t *raw = nullptr;  // raw_data span inside the file buffer (may be null)
    size_t raw_bytes = 0;
    std::vector<float> f32;        // float_data (when raw is absent)
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
    // float view (nullptr if not float32, or if the file holds fewer values than the dims claim)
    const float *data() const {
        if (dtype != 1 || !consistent()) return nullptr;
        if (raw) return reinterpret_cast<const float *>(raw);
        return f32.empty() ? nullptr : f32.data();
    }
    // do the dims describe exactly what the file stores?  (damaged files: never trust a length field)
    bool consistent() const {
        int64_t n = 1;
        for (auto d : dims) {
            if (d < 0 || d > (int64_t(1) << 31)) return false;
            n *= d;
            if (n > (int64_t(1) << 33)) return false;
        }
        const size_t have = raw ? raw_bytes / 4 : f32.size();
        return size_t(n) <= have;
    }
};

struct OnnxNode {
    std::string name, op;
    std::vector<std::string> inputs, outputs;
    std::map<std::string, std::vector<int64_t>> ints;  // attribute ints / single i
    std::map<std::string, float> floats;               // attribute f (LayerNormalization epsilon)
};

struct OnnxModel {
    std::vector<uint8_t> buf;  // whole file
    std::vector<OnnxNode> nodes;
    std::map<std::string, OnnxTensor> init;
    // Identity(initializer) outputs: the exporter de-duplicates initializers with identical bytes and re-introduces the
    // other names this way (alias -> kept name)
    std::map<std::string, std::string> alias;
    // initializer by name, following aliases; nullptr if absent
    const OnnxTensor *find_init(const std::string &name) const {
        auto it = init.find(name);
        if (it != init.end()) return &it->second;
        std::string cur = name;
        for (int hop = 0; hop < 8; hop++) {
            auto a = alias.find(cur);
            if (a == alias.end()) return nullptr;
            cur = a->second;
            it = init.find(cur);
            if (it != init.end()) return &it->second;
        }
        return nullptr;
    }
    std::vector<std::string> inputs;  // graph inputs that are not initializers
    std::vector<std::string> outputs;
    std::map<std::string, std::string> meta;
    int64_t opset = 0;
    // returns empty string on success, else the error
    std::string load(const std::string &path);
};

}  // namespace vitsmi
