// Hand-written ONNX protobuf reader (wire format only; no protoc, no onnx package offline).
// Replaces what the reference delegates to nvonnxparser::IParser::parseFromFile
// (/root/reference/src/tensorrt/img2img_build.cpp:81-88).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace w2x {

// ONNX TensorProto.DataType values we understand.
enum DType : int { DT_F32 = 1, DT_U8 = 2, DT_I8 = 3, DT_I32 = 6, DT_I64 = 7, DT_BOOL = 9, DT_F16 = 10, DT_F64 = 11 };

// Host tensor: integers/bools live in `i`, floating point in `f` (fp16/fp64 initialisers are widened/narrowed to fp32).
struct HTensor {
    int dtype = DT_F32;
    std::vector<int64_t> shape;
    std::vector<int64_t> i;
    std::vector<float> f;
    bool is_float() const { return dtype == DT_F32 || dtype == DT_F16 || dtype == DT_F64; }
    int64_t numel() const { int64_t n = 1; for (auto d : shape) n *= d; return n; }
    int rank() const { return (int)shape.size(); }
};
using HTensorP = std::shared_ptr<HTensor>;

struct Attr {
    int type = 0;  // 1 f, 2 i, 3 s, 4 t, 6 floats, 7 ints
    float f = 0;
    int64_t i = 0;
    std::string s;
    HTensorP t;
    std::vector<float> floats;
    std::vector<int64_t> ints;
};

struct Node {
    std::string op, name;
    std::vector<std::string> in, out;
    std::map<std::string, Attr> attr;
    int64_t ai(const std::string& k, int64_t d) const { auto it = attr.find(k); return it == attr.end() ? d : it->second.i; }
    float af(const std::string& k, float d) const { auto it = attr.find(k); return it == attr.end() ? d : it->second.f; }
    std::vector<int64_t> aints(const std::string& k) const { auto it = attr.find(k); return it == attr.end() ? std::vector<int64_t>{} : it->second.ints; }
    bool has(const std::string& k) const { return attr.count(k) != 0; }
};

struct ValueInfo {
    std::string name;
    int elem_type = 0;
    std::vector<int64_t> dims;  // -1 = dynamic (dim_param)
};

struct Model {
    int64_t ir_version = 0, opset = 0;
    std::string producer;
    std::vector<Node> nodes;
    std::map<std::string, HTensorP> init;
    std::vector<ValueInfo> inputs, outputs;
};

// Throws std::runtime_error on malformed input.
Model load_onnx(const std::string& path);

}  // namespace w2x
