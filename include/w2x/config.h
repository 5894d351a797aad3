// Configuration structs of the engine interface.
// Mirrors /root/reference/src/tensorrt/config.h:7-43 (trt::Precision, trt::BuildConfig, trt::RenderConfig):
// same fields, same defaults.  cv::Point2d overlap becomes two doubles (OpenCV is not a dependency here).
#ifndef W2X_CONFIG_H
#define W2X_CONFIG_H

namespace w2x {

enum class Precision {
    TF32,   // config.h:8.  gfx950 has no TF32/xf32 MFMA: accepted by the parser, rejected by build()/load() with a message.
    FP16    // config.h:9
};

struct BuildConfig {      // config.h:12-31
    int deviceId = 0;
    Precision precision = Precision::FP16;

    int minBatchSize = 1;
    int optBatchSize = 1;
    int maxBatchSize = 4;

    int minChannels = 3;
    int optChannels = 3;
    int maxChannels = 3;

    int minWidth = 64;
    int optWidth = 256;
    int maxWidth = 640;

    int minHeight = 64;
    int optHeight = 256;
    int maxHeight = 640;
};

struct RenderConfig {     // config.h:33-43
    int deviceId = 0;
    Precision precision = Precision::FP16;
    int batchSize = 1;
    int channels = 3;
    int height = 256;
    int width = 256;
    int scaling = 4;
    double overlapX = 0.0625;   // cv::Point2d overlap = (0.0625, 0.0625)
    double overlapY = 0.0625;
    bool tta = false;
    // extension (not in the reference): reproduce quirk Q1 (img2img_render.cpp:313-316, the TTA mean is computed and
    // then the last de-augmented output is blended instead).  Default false = the intended mean.
    bool ttaBugCompat = false;
};

}  // namespace w2x

#endif
