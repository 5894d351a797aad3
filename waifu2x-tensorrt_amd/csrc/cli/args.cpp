#include "args.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <filesystem>
#include <sstream>
#include <stdexcept>

namespace w2x::cli {

namespace {

int to_int(const std::string& name, const std::string& v) {
    char* end = nullptr;
    long x = std::strtol(v.c_str(), &end, 10);
    if (v.empty() || *end) throw std::runtime_error(name + ": '" + v + "' is not an integer");
    return (int)x;
}

double to_double(const std::string& name, const std::string& v) {
    // accepts decimals and simple fractions ("1/16"), the way the README writes the blend choices
    auto slash = v.find('/');
    char* end = nullptr;
    if (slash != std::string::npos) {
        double a = std::strtod(v.substr(0, slash).c_str(), &end), b = std::strtod(v.substr(slash + 1).c_str(), nullptr);
        if (b == 0) throw std::runtime_error(name + ": bad fraction '" + v + "'");
        return a / b;
    }
    double x = std::strtod(v.c_str(), &end);
    if (v.empty() || *end) throw std::runtime_error(name + ": '" + v + "' is not a number");
    return x;
}

template <class T> void member(const std::string& name, const T& v, std::initializer_list<T> set) {
    if (std::find(set.begin(), set.end(), v) == set.end()) {
        std::ostringstream os; os << name << ": " << v << " not in {";
        bool first = true; for (const T& s : set) { os << (first ? "" : ",") << s; first = false; }
        os << "}";
        throw std::runtime_error(os.str());
    }
}

}  // namespace

std::string usage() {
    return "waifu2x (MI355X-native) - same command line as z3lx/waifu2x-tensorrt\n"
           "Usage: w2x [OPTIONS] SUBCOMMAND\n\n"
           "Options:\n"
           "  -h,--help                   Print this help message and exit\n"
           "  --model TEXT REQUIRED       {cunet/art,swin_unet/art,swin_unet/art_scan,swin_unet/photo}\n"
           "  --scale INT REQUIRED        {1,2,4}\n"
           "  --noise INT REQUIRED        {-1,0,1,2,3}\n"
           "  --batchSize INT REQUIRED    > 0\n"
           "  --tileSize INT REQUIRED     {64,128,256,400,640}\n"
           "  --device INT [0]            GPU device ID\n"
           "  --precision TEXT [fp16]     {fp16,tf32,fp32}\n"
           "  --devices INT [1]           (extension) number of GPUs: video frames round-robin, one image as N tile ranges\n"
           "  --split TEXT [shards]       (extension) {shards,strips}: one image over --devices N: every tile once with the seam bands exchanged / whole tile columns\n"
           "  --deep                      (extension) 16-bit PNGs keep 16 bits per sample through the engine and in the output\n"
           "  --models DIR [models]       (extension) root of the model directory tree\n\n"
           "Subcommands:\n"
           "  render                      Render image(s)/video(s)\n"
           "      -i,--input PATH ... REQUIRED   --recursive   -o,--output DIR   --nosuffix\n"
           "      --blend FLOAT [1/16] {1/8,1/16,1/32,0}   --tta   --codec TEXT [libx264]   --pix_fmt TEXT [yuv420p]   --crf INT [23] 0..51\n"
           "      --tta-mode TEXT [mean]  (extension) {mean,reference}: with --tta, `mean` averages the 8 augmentations; `reference`\n"
           "                              reproduces the bytes of the reference's accumulation (img2img_render.cpp:313-316); --tta-compat = reference\n"
           "  build                       Build model\n"
           "  convert -i IN -o OUT        (extension) re-encode one still image (png/ppm), no GPU\n";
}

Options parse(int argc, const char* const* argv) {
    Options o;
    std::vector<std::string> a(argv + 1, argv + argc);
    bool seen_model = false, seen_scale = false, seen_noise = false, seen_batch = false, seen_tile = false;
    auto value = [&](size_t& i) -> std::string {
        const std::string name = a[i];
        auto eq = name.find('=');
        if (name.rfind("--", 0) == 0 && eq != std::string::npos) { std::string v = name.substr(eq + 1); a[i] = name.substr(0, eq); return v; }
        if (i + 1 >= a.size()) throw std::runtime_error(name + ": 1 required value missing");
        return a[++i];
    };
    for (size_t i = 0; i < a.size(); ++i) {
        std::string k = a[i];
        if (k.rfind("--", 0) == 0 && k.find('=') != std::string::npos) k = k.substr(0, k.find('='));
        if (k == "-h" || k == "--help") { o.help = true; return o; }
        else if (k == "render" || k == "build" || k == "convert") {
            if (!o.command.empty()) throw std::runtime_error("Exactly 1 subcommand is required");
            o.command = k;
        }
        else if (k == "--model") { o.model = value(i); seen_model = true; }
        else if (k == "--scale") { o.scale = to_int(k, value(i)); seen_scale = true; }
        else if (k == "--noise") { o.noise = to_int(k, value(i)); seen_noise = true; }
        else if (k == "--batchSize") { o.batchSize = to_int(k, value(i)); seen_batch = true; }
        else if (k == "--tileSize") { o.tileSize = to_int(k, value(i)); seen_tile = true; }
        else if (k == "--device") o.device = to_int(k, value(i));
        else if (k == "--devices") o.devices = to_int(k, value(i));
        else if (k == "--split") o.split = value(i);
        else if (k == "--models") o.models = value(i);
        else if (k == "--precision") { o.precision = value(i); std::transform(o.precision.begin(), o.precision.end(), o.precision.begin(), ::tolower); }
        else if (k == "--print-config") o.printConfig = true;
        else if (k == "--deep") o.deep = true;
        else if (k == "-i" || k == "--input") {
            o.inputs.push_back(value(i));
            while (i + 1 < a.size() && a[i + 1].rfind("-", 0) != 0 && a[i + 1] != "render" && a[i + 1] != "build" && a[i + 1] != "convert") o.inputs.push_back(a[++i]);
        }
        else if (k == "--recursive") o.recursive = true;
        else if (k == "-o" || k == "--output") o.output = value(i);
        else if (k == "--nosuffix") o.nosuffix = true;
        else if (k == "--blend") o.blend = to_double(k, value(i));
        else if (k == "--tta") o.tta = true;
        else if (k == "--tta-mode") { o.ttaMode = value(i); std::transform(o.ttaMode.begin(), o.ttaMode.end(), o.ttaMode.begin(), ::tolower); }
        else if (k == "--tta-compat") o.ttaMode = "reference";
        else if (k == "--codec") o.codec = value(i);
        else if (k == "--pix_fmt") o.pixFmt = value(i);
        else if (k == "--crf") o.crf = to_int(k, value(i));
        else throw std::runtime_error("The following argument was not expected: " + a[i]);
    }
    if (o.command.empty()) throw std::runtime_error("A subcommand is required");
    if (o.command == "convert") {
        if (o.inputs.size() != 1 || o.output.empty()) throw std::runtime_error("convert: exactly one -i and one -o are required");
        return o;
    }
    if (!seen_model) throw std::runtime_error("--model is required");
    if (!seen_scale) throw std::runtime_error("--scale is required");
    if (!seen_noise) throw std::runtime_error("--noise is required");
    if (!seen_batch) throw std::runtime_error("--batchSize is required");
    if (!seen_tile) throw std::runtime_error("--tileSize is required");
    member<std::string>("--model", o.model, {"cunet/art", "swin_unet/art", "swin_unet/art_scan", "swin_unet/photo"});
    member("--scale", o.scale, {1, 2, 4});
    member("--noise", o.noise, {-1, 0, 1, 2, 3});
    if (o.batchSize <= 0) throw std::runtime_error("--batchSize: number must be positive");
    member("--tileSize", o.tileSize, {64, 128, 256, 400, 640});
    if (o.device < 0) throw std::runtime_error("--device: number must be non-negative");
    if (o.devices < 1) throw std::runtime_error("--devices: number must be positive");
    member<std::string>("--split", o.split, {"shards", "strips"});
    member<std::string>("--precision", o.precision, {"fp16", "tf32", "fp32"});   // fp32: an addition (include/w2x/config.h)
    if (o.command == "render") {
        if (o.inputs.empty()) throw std::runtime_error("--input is required");
        for (const auto& p : o.inputs) if (!std::filesystem::exists(p)) throw std::runtime_error("--input: Path does not exist: " + p);
        if (!o.output.empty() && !std::filesystem::is_directory(o.output)) throw std::runtime_error("--output: Directory does not exist: " + o.output);
        const double choices[] = {1.0 / 8.0, 1.0 / 16.0, 1.0 / 32.0, 0.0};
        if (std::none_of(std::begin(choices), std::end(choices), [&](double c) { return c == o.blend; })) throw std::runtime_error("--blend: not in {1/8,1/16,1/32,0}");
        if (o.crf < 0 || o.crf > 51) throw std::runtime_error("--crf: Value not in range 0 to 51");
        member<std::string>("--tta-mode", o.ttaMode, {"mean", "reference"});
        if (o.ttaMode == "reference" && !o.tta) throw std::runtime_error("--tta-mode reference: needs --tta");
    }
    // cross-checks, main.cpp:142-145
    if (o.model == "cunet/art" && o.scale == 4) throw std::runtime_error("cunet/art does not support scale factor 4.");
    if (o.noise == -1 && o.scale == 1) throw std::runtime_error("Noise level -1 does not support scale factor 1.");
    return o;
}

std::string model_path(const Options& o) {
    return o.models + "/" + o.model + "/" + (o.noise == -1 ? "" : "noise" + std::to_string(o.noise) + "_") +
           (o.scale == 1 ? "" : "scale" + std::to_string(o.scale) + "x") + ".onnx";
}

std::string output_suffix(const Options& o) {
    std::string m = o.model;
    std::replace(m.begin(), m.end(), '/', '_');
    return "(" + m + ")" + (o.noise == -1 ? "" : "(noise" + std::to_string(o.noise) + ")") +
           (o.scale == 1 ? "" : "(scale" + std::to_string(o.scale) + ")") + (o.tta ? "(tta)" : "");
}

std::string output_path(const Options& o, const std::string& input, bool single_frame) {
    namespace fs = std::filesystem;
    fs::path file(input);
    if (!o.output.empty()) file = fs::path(o.output) / file.filename();
    if (!o.nosuffix) file.replace_filename(file.stem().string() + output_suffix(o) + file.extension().string());
    file.replace_extension(single_frame ? ".png" : ".mp4");
    return file.string();
}

std::string to_json(const Options& o) {
    auto q = [](const std::string& s) { std::string r = "\""; for (char c : s) { if (c == '"' || c == '\\') r += '\\'; r += c; } return r + "\""; };
    std::ostringstream os;
    os << "{\"command\": " << q(o.command) << ", \"model\": " << q(o.model) << ", \"scale\": " << o.scale << ", \"noise\": " << o.noise
       << ", \"batchSize\": " << o.batchSize << ", \"tileSize\": " << o.tileSize << ", \"device\": " << o.device << ", \"devices\": " << o.devices << ", \"split\": " << q(o.split)
       << ", \"precision\": " << q(o.precision) << ", \"recursive\": " << (o.recursive ? "true" : "false") << ", \"output\": " << q(o.output)
       << ", \"nosuffix\": " << (o.nosuffix ? "true" : "false") << ", \"blend\": " << o.blend << ", \"tta\": " << (o.tta ? "true" : "false") << ", \"tta_mode\": " << q(o.ttaMode)
       << ", \"codec\": " << q(o.codec) << ", \"pix_fmt\": " << q(o.pixFmt) << ", \"crf\": " << o.crf << ", \"inputs\": [";
    for (size_t i = 0; i < o.inputs.size(); ++i) os << (i ? ", " : "") << q(o.inputs[i]);
    os << "], \"model_path\": " << q(o.command == "convert" ? "" : model_path(o)) << ", \"suffix\": " << q(o.command == "convert" ? "" : output_suffix(o)) << ", \"outputs\": [";
    if (o.command == "render") for (size_t i = 0; i < o.inputs.size(); ++i) os << (i ? ", " : "") << q(output_path(o, o.inputs[i], true));
    os << "]}";
    return os.str();
}

}  // namespace w2x::cli
