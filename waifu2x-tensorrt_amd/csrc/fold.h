// Static-shape specialisation of an ONNX graph: constant folding of everything that does not depend on the
// runtime input, plus shape inference for what does.  This is the job TensorRT's builder performs when the
// reference pins min=opt=max shapes (/root/reference/src/main.cpp:276-291, img2img_build.cpp:102-116).
#pragma once
#include <deque>

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
    std::vector<const Node*> nodes;            // nodes with at least one runtime-dependent input, in canonical (depth-first from the output) order
    std::deque<Node> owned;                    // the nodes above: copies of the model's, rewritten by simplify_graph (simplify.cpp)
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

// Canonical spelling of the runtime part of a folded graph (simplify.cpp; fold_graph ends with it): no-op nodes removed, Gemm as MatMul + Add,
// Flatten / Squeeze / Unsqueeze as Reshape, chains of Transposes / Reshapes merged, nodes in an order that depends on the graph alone.
void simplify_graph(FoldedGraph& g);

}  // namespace w2x
