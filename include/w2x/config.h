// Configuration structs of the engine interface: the members, member order and defaults of trt::Precision,
// trt::BuildConfig and trt::RenderConfig (/root/reference/src/tensorrt/config.h:7-43), so that code written against the
// reference's structs compiles against these.  cv::Point2d overlap becomes two doubles (OpenCV is not a dependency here).
#ifndef W2X_CONFIG_H
#define W2X_CONFIG_H

namespace w2x {

// Arithmetic of the network.  TF32 and FP16 are the reference's two values, with their values.  gfx950 has no TF32 / xf32 matrix
// instruction, so a Precision::TF32 request gets the fp32-storage engine (fp32 maps and weights, fp32 accumulation, the un-fused
// operator set) with TF32-CLASS products: each operand is split into two bf16 halves and a product is three v_mfma_f32_16x16x32_bf16
// (16 significant bits per operand where TF32 keeps 11; the bf16 exponent is fp32's, so nothing can overflow that fp32 holds).
// FP32 (not in the reference) is the same engine with exact fp32 products on v_mfma_f32_16x16x4_f32: about twice the frame time,
// outputs within summation order of an fp32 evaluation of the graph.  An engine file serves only the precision it was built for
// (img2img_load.cpp:54-66).
enum class Precision { TF32, FP16, FP32 };

// What build() specialises a plan for.  The reference hands TensorRT a min / opt / max optimisation profile per
// dimension (img2img_build.cpp:102-116); its command line always sets the three equal (main.cpp:276-291).  build() checks
// min <= opt <= max, writes the plan for the opt shape and keys the plan cache on all of them, like the reference's JSON side
// file; load() takes the first optimized engine, else the first compatible one (img2img_load.cpp:100-107) and re-specialises
// the plan from the same ONNX file for a shape inside the range that is not the opt shape.
struct BuildConfig {
    int deviceId = 0;                                             // HIP device ordinal
    Precision precision = Precision::FP16;
    int minBatchSize = 1, optBatchSize = 1, maxBatchSize = 4;     // tiles per network call
    int minChannels = 3, optChannels = 3, maxChannels = 3;        // always RGB
    int minWidth = 64, optWidth = 256, maxWidth = 640;            // tile width in pixels
    int minHeight = 64, optHeight = 256, maxHeight = 640;         // tile height in pixels
};

// What load() prepares render() for: one plan (batch, tile size), the scale the caller expects, the blend overlap as a
// fraction of the tile, and test-time augmentation.
struct RenderConfig {
    int deviceId = 0;
    Precision precision = Precision::FP16;
    int batchSize = 1;
    int channels = 3;
    int height = 256, width = 256;                                // tile size
    int scaling = 4;                                              // comes from the caller, not from the graph
    double overlapX = 0.0625, overlapY = 0.0625;                  // the reference's cv::Point2d overlap
    bool tta = false;
    // Not in the reference: reproduce its quirk Q1 (img2img_render.cpp:313-316 computes the TTA mean and then blends the
    // last de-augmented output instead).  false = the intended mean.
    bool ttaBugCompat = false;
};

}  // namespace w2x

#endif
