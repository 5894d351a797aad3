// Static-shape specialisation of an ONNX graph: constant folding of everything that does not depend on the
// runtime input, plus shape inference for what does.  This is the job TensorRT's builder performs when the
// reference pins min=opt=max shapes (/root/reference/src/main.cpp:276-291, img2img_build.cpp:102-116).
#pragma once
#include "onnx_pb.h"

namespace w2x {

struct Value {
    bool is_const = false;
    HTensorP c;                  // when is_const
    std::vector<int64_t> shape;  // always set
    int dtype = DT_F32;
};

struct FoldedGraph {
    const Model* model = nullptr;
    std::vector<const Node*> nodes;            // nodes with at least one runtime-dependent input, topological order
    std::map<std::string, Value> vals;         // every tensor name -> value/shape
    std::map<std::string, std::vector<const Node*>> consumers;  // among `nodes`
    std::map<std::string, const Node*> producer;                // among `nodes`
    std::string input, output;
    const Value& val(const std::string& n) const;
    bool is_const(const std::string& n) const { auto it = vals.find(n); return it != vals.end() && it->second.is_const; }
    const HTensor& cst(const std::string& n) const;
};

// Throws std::runtime_error (unsupported op during folding/shape inference, inconsistent shapes, ...).
FoldedGraph fold_graph(const Model& m, const std::vector<int64_t>& input_shape);

}  // namespace w2x
