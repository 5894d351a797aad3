// ONNX (folded to static shapes) -> Plan.  The graph is walked in topological order and matched against the
// fused operator set the HIP kernels implement; anything unmatched fails the build with the offending node named,
// the way TensorRT's parser rejects unsupported layers (/root/reference/src/tensorrt/img2img_build.cpp:81-88).
#pragma once
#include "fold.h"
#include "plan.h"

namespace w2x {

// fp32: activations, weights and bias tables in fp32 and none of the fused (fp16) transformer kernels - the engine's second precision
Plan lower_graph(const FoldedGraph& g, bool fp32 = false);

// Convenience: load + fold + lower.  input is [B,3,T,T].
Plan build_plan(const std::string& onnx_path, int batch, int channels, int height, int width, bool fp32 = false);
std::string onnx_op_histogram(const std::string& onnx_path);

}  // namespace w2x
