// Command line of the reference (src/main.cpp:18-153): global model options, exactly one of the subcommands
// `render` / `build`, options may appear before or after the subcommand (CLI11 "fallthrough").  Hand-written parser:
// no CLI11 in this tree.  Extensions are marked below.
#pragma once
#include <string>
#include <vector>

namespace w2x::cli {

struct Options {
    // global (main.cpp:25-84)
    std::string model;                    // cunet/art | swin_unet/art | swin_unet/art_scan | swin_unet/photo
    int scale = 0, noise = -2, batchSize = 0, tileSize = 0;
    int device = 0;
    std::string precision = "fp16";       // fp16 | tf32 (tf32 is rejected by build/load on gfx950, like platformHasTf32()==false)
    // subcommand
    std::string command;                  // "render" | "build" | "convert" (extension: image format conversion, no GPU)
    // render (main.cpp:86-139)
    std::vector<std::string> inputs;
    bool recursive = false;
    std::string output;                   // directory
    bool nosuffix = false;
    double blend = 1.0 / 16.0;
    bool tta = false;
    std::string codec = "libx264", pixFmt = "yuv420p";
    int crf = 23;
    // extensions
    int devices = 1;                      // --devices N: frames of a video round-robin, a single image as N tile ranges (--split)
    std::string split = "shards";         // --split {shards,strips}: how ONE image is spread over --devices N: shards = every tile once, seam bands exchanged
                                          // (Img2Img::renderSharded); strips = whole tile columns per device, the seam column recomputed (Img2Img::renderStrip)
    std::string models = "models";        // --models DIR: root of models/<model>/... (reference: fixed relative "models/")
    std::string ttaMode = "mean";         // --tta-mode {mean,reference}: mean = the true average of the 8 augmentations; reference = the bytes the
                                          // reference's accumulation produces (img2img_render.cpp:313-316) -> RenderConfig::ttaBugCompat
    bool deep = false;                    // --deep: 16-bit PNGs keep 16 bits per sample (read, rendered and written as CV_16UC3; default: cut to 8 like cv::imread)
    bool printConfig = false;             // --print-config: dump the parsed options and derived names as JSON and exit (tests)
    bool help = false;
};

// throws std::runtime_error with the message to print (exit code -1 like main.cpp:147-150) on invalid input
Options parse(int argc, const char* const* argv);
std::string usage();

// models/<model>/[noiseN_][scaleSx].onnx  (main.cpp:201-204)
std::string model_path(const Options& o);
// "(model_with_underscores)(noiseN)(scaleS)(tta)"  (main.cpp:205-209)
std::string output_suffix(const Options& o);
// output file name for one input (main.cpp:240-257): directory override, suffix, .png for stills / .mp4 for videos
std::string output_path(const Options& o, const std::string& input, bool single_frame);
std::string to_json(const Options& o);

}  // namespace w2x::cli
