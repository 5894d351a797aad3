// w2x: the reference's command line (src/main.cpp) on top of w2x::Img2Img.
//   w2x --model swin_unet/art --scale 4 --noise 3 --batchSize 4 --tileSize 256 build
//   w2x --model swin_unet/art --scale 4 --noise 3 --batchSize 4 --tileSize 256 render -i in.png -o outdir [--tta] [--blend 1/16]
// Stills (.png / .ppm / .bmp) and uncompressed .avi files are read and written by the built-in codecs (imageio.h); other formats and
// videos are piped through ffmpeg as raw bgr24 when ffmpeg/ffprobe are on PATH (videoio/capture.cpp:96-99, writer.cpp:24-33 do the same).
// A PNG / BMP with an alpha channel keeps it: the alpha plane goes through the same engine as a gray image (upstream TODO, README.md:88).
// Extension: --devices N drives N engines - a single image is split into tile-column strips (Img2Img::renderStrip), every engine
// writing its own columns of the shared output buffer; a video is cut into chunks of frames that go round-robin to one persistent
// worker thread per engine (renderSequence over that engine's page-locked buffers), with one reader and one in-order writer thread.
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <atomic>
#include <condition_variable>
#include <csignal>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "../../../include/w2x/img2img.h"
#include "args.h"
#include "imageio.h"

using namespace w2x;
namespace fs = std::filesystem;

namespace {

bool on_path(const char* tool) { return std::system((std::string("command -v ") + tool + " >/dev/null 2>&1").c_str()) == 0; }

std::string shell_quote(const std::string& s) { std::string r = "'"; for (char c : s) { if (c == '\'') r += "'\\''"; else r += c; } return r + "'"; }

struct Probe { int width = 0, height = 0, frames = 0; double fps = 0; };

Probe ffprobe(const std::string& file) {
    const std::string cmd = "ffprobe -v error -select_streams v:0 -count_packets -show_entries stream=width,height,r_frame_rate,nb_read_packets -of csv=p=0 " + shell_quote(file);
    Probe p;
    if (FILE* f = popen(cmd.c_str(), "r")) {
        int num = 0, den = 1;
        if (fscanf(f, "%d,%d,%d/%d,%d", &p.width, &p.height, &num, &den, &p.frames) >= 4 && den) p.fps = (double)num / den;
        pclose(f);
    }
    if (p.width <= 0 || p.height <= 0) throw std::runtime_error("ffprobe could not read " + file);
    if (p.frames <= 0) p.frames = 1;
    return p;
}

std::vector<std::string> find_inputs(const cli::Options& o) {
    static const char* exts[] = {".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".ppm", ".mp4", ".avi", ".mkv"};   // main.cpp:156-159 (+ .ppm)
    auto wanted = [&](const fs::path& p) { std::string e = p.extension().string(); for (auto& c : e) c = (char)tolower(c); for (const char* x : exts) if (e == x) return true; return false; };
    std::vector<std::string> files;
    for (const std::string& in : o.inputs) {
        if (fs::is_directory(in)) {
            if (o.recursive) { for (const auto& e : fs::recursive_directory_iterator(in)) if (e.is_regular_file() && wanted(e.path())) files.push_back(e.path().string()); }
            else for (const auto& e : fs::directory_iterator(in)) if (e.is_regular_file() && wanted(e.path())) files.push_back(e.path().string());
        } else if (wanted(in)) files.push_back(in);
    }
    std::sort(files.begin(), files.end());
    return files;
}

// ---- frame streams: raw bgr24 frames from / to an ffmpeg pipe or a built-in uncompressed AVI
struct FrameSource { virtual ~FrameSource() {} virtual bool read(uint8_t* bgr) = 0; };
struct FrameSink { virtual ~FrameSink() {} virtual bool write(const uint8_t* bgr) = 0; virtual bool close() = 0; };

struct PipeSource : FrameSource {
    FILE* f; size_t bytes;
    PipeSource(const std::string& cmd, size_t b) : f(popen(cmd.c_str(), "r")), bytes(b) { if (!f) throw std::runtime_error("cannot start ffmpeg"); }
    ~PipeSource() override { if (f) pclose(f); }
    bool read(uint8_t* bgr) override { return fread(bgr, 1, bytes, f) == bytes; }
};
struct PipeSink : FrameSink {
    FILE* f; size_t bytes;
    PipeSink(const std::string& cmd, size_t b) : f(popen(cmd.c_str(), "w")), bytes(b) { if (!f) throw std::runtime_error("cannot start ffmpeg"); }
    ~PipeSink() override { if (f) pclose(f); }
    bool write(const uint8_t* bgr) override { return fwrite(bgr, 1, bytes, f) == bytes; }
    bool close() override { FILE* g = f; f = nullptr; return g && pclose(g) == 0; }
};
struct AviSource : FrameSource {
    cli::AviReader rd;
    bool read(uint8_t* bgr) override { return rd.read(bgr); }
};
struct AviSink : FrameSink {
    cli::AviWriter wr;
    bool write(const uint8_t* bgr) override { try { wr.write(bgr); return true; } catch (const std::exception& e) { std::cerr << e.what() << "\n"; return false; } }
    bool close() override { try { wr.close(); return true; } catch (const std::exception& e) { std::cerr << e.what() << "\n"; return false; } }
};

// The frame loop of main.cpp:263-269 over N engines.  Chunk c (CH consecutive frames) belongs to engine c % N; every engine owns SLOTS
// chunk slots of page-locked frame buffers (allocHost: the copies of renderSequence run by DMA beside the kernels).  One reader thread
// fills slots in stream order, one persistent worker per engine renders its chunks with renderSequence (upload / compute / download of
// consecutive frames overlapped), one writer thread drains the slots in stream order - decoding, N renders and encoding all overlap,
// and frames leave in the order they came (the reference serialises read -> render -> write per frame).
bool run_frame_pipeline(std::vector<std::unique_ptr<Img2Img>>& engines, FrameSource& src, FrameSink& dst, int width, int height, int scale,
                        const std::function<void(int)>& on_frames) {
    const int N = (int)engines.size(), CH = 4, SLOTS = 2;
    const size_t inBytes = (size_t)width * height * 3, outBytes = inBytes * scale * scale;
    struct Slot { std::vector<uint8_t*> in, out; int frames = 0; int state = 0; };   // 0 free, 1 read, 2 rendered
    std::vector<std::vector<Slot>> slots(N, std::vector<Slot>(SLOTS));
    auto release = [&] { for (int e = 0; e < N; ++e) for (Slot& sl : slots[e]) { for (uint8_t* p : sl.in) engines[e]->freeHost(p); for (uint8_t* p : sl.out) engines[e]->freeHost(p); sl.in.clear(); sl.out.clear(); } };
    for (int e = 0; e < N; ++e)
        for (Slot& sl : slots[e])
            for (int k = 0; k < CH; ++k) {
                sl.in.push_back((uint8_t*)engines[e]->allocHost(inBytes)); sl.out.push_back((uint8_t*)engines[e]->allocHost(outBytes));
                if (!sl.in.back() || !sl.out.back()) { release(); throw std::runtime_error("cannot allocate page-locked frame buffers"); }
            }
    std::mutex mu; std::condition_variable cv;
    bool failed = false;
    long total_chunks = -1;                         // known once the reader has hit the end of the stream
    auto slot_of = [&](long c) -> Slot& { return slots[c % N][(c / N) % SLOTS]; };
    std::thread reader([&] {
        for (long c = 0;; ++c) {
            Slot& sl = slot_of(c);
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sl.state == 0 || failed; }); if (failed) return; }
            int got = 0;
            for (; got < CH; ++got) if (!src.read(sl.in[got])) break;
            { std::lock_guard<std::mutex> lk(mu); sl.frames = got; if (got) sl.state = 1; if (got < CH) total_chunks = c + (got ? 1 : 0); }
            cv.notify_all();
            if (got < CH) return;
        }
    });
    std::vector<std::thread> workers;
    for (int e = 0; e < N; ++e) workers.emplace_back([&, e] {
        for (long c = e;; c += N) {
            Slot& sl = slot_of(c);
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sl.state == 1 || failed || (total_chunks >= 0 && c >= total_chunks); }); if (failed || sl.state != 1) return; }
            const int got = sl.frames;
            std::vector<Image> si(got), di(got);
            for (int k = 0; k < got; ++k) {
                si[k] = Image{sl.in[k], height, width, (size_t)width * 3};
                di[k] = Image{sl.out[k], height * scale, width * scale, (size_t)width * scale * 3};
            }
            const bool ok = engines[e]->renderSequence(si.data(), di.data(), got);
            { std::lock_guard<std::mutex> lk(mu); if (!ok) failed = true; sl.state = 2; }
            cv.notify_all();
            if (!ok) return;
        }
    });
    std::thread writer([&] {
        for (long c = 0;; ++c) {
            Slot& sl = slot_of(c);
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sl.state == 2 || failed || (total_chunks >= 0 && c >= total_chunks); }); if (failed || sl.state != 2) return; }
            const int n = sl.frames;
            for (int k = 0; k < n; ++k) if (!dst.write(sl.out[k])) { std::lock_guard<std::mutex> lk(mu); failed = true; }
            { std::lock_guard<std::mutex> lk(mu); sl.state = 0; }
            cv.notify_all();
            on_frames(n);
        }
    });
    reader.join();
    for (auto& t : workers) t.join();
    writer.join();
    release();
    return !failed;
}

}  // namespace

int main(int argc, char** argv) {
    cli::Options o;
    try { o = cli::parse(argc, argv); }
    catch (const std::exception& e) { std::cerr << e.what() << "\nRun with --help for more information.\n"; return -1; }
    if (o.help) { std::cout << cli::usage(); return 0; }
    if (o.printConfig) { std::cout << cli::to_json(o) << "\n"; return 0; }
    try {
        if (o.command == "convert") { cli::write_image(o.output, cli::read_image(o.inputs[0])); return 0; }

        const std::string modelPath = cli::model_path(o);
        // the progress callback runs on the engines' worker threads while the writer thread advances the frame counter
        std::atomic<size_t> fileIndex{0}, fileCount{0}, frameIndex{0}, frameCount{0};
        auto on_message = [](Severity s, const std::string& m) {
            static const char* names[] = {"critical", "error", "warning", "info", "debug", "trace"};
            std::cerr << "[" << names[(int)s < 6 ? (int)s : 5] << "] " << m << "\n";
        };
        auto on_progress = [&](int current, int total, double speed) {
            fprintf(stderr, "[info] Rendered file %zu/%zu, frame %zu/%zu, batch %d/%d @ %.2f it/s\n", fileIndex.load(), fileCount.load(), frameIndex.load(), frameCount.load(), current, total, speed);
        };
        const Precision prec = o.precision == "tf32" ? Precision::TF32 : o.precision == "fp32" ? Precision::FP32 : Precision::FP16;

        if (o.command == "build") {
            Img2Img engine;
            engine.setMessageCallback(on_message);
            BuildConfig c;
            c.deviceId = o.device; c.precision = prec;
            c.minBatchSize = c.optBatchSize = c.maxBatchSize = o.batchSize;
            c.minChannels = c.optChannels = c.maxChannels = 3;                       // main.cpp:280-282
            c.minWidth = c.optWidth = c.maxWidth = c.minHeight = c.optHeight = c.maxHeight = o.tileSize;
            return engine.build(modelPath, c) ? 0 : -1;
        }

        // render: one engine per device
        std::vector<std::unique_ptr<Img2Img>> engines;
        // external tools are probed (fork + exec of a shell) before the first engine brings the HIP runtime up; a dying encoder must
        // surface as a failed write, not as SIGPIPE
        const bool have_ffmpeg = on_path("ffmpeg") && on_path("ffprobe");
        signal(SIGPIPE, SIG_IGN);
        for (int d = 0; d < o.devices; ++d) {
            engines.emplace_back(new Img2Img);
            engines.back()->setMessageCallback(on_message);
            if (d == 0) engines.back()->setProgressCallback(on_progress);
            RenderConfig c;
            c.deviceId = o.device + d; c.precision = prec; c.batchSize = o.batchSize; c.channels = 3; c.height = c.width = o.tileSize; c.scaling = o.scale;
            c.overlapX = c.overlapY = o.blend; c.tta = o.tta; c.ttaBugCompat = o.ttaMode == "reference";
            if (!engines.back()->load(modelPath, c)) return -1;
        }
        const std::vector<std::string> files = find_inputs(o);
        fileCount = files.size();
        for (const std::string& file : files) {
            if (cli::is_builtin_still(file)) {
                frameIndex = 0; frameCount = 1;
                cli::Bitmap in = cli::read_image(file, o.deep), out;
                out.rows = in.rows * o.scale; out.cols = in.cols * o.scale;
                const bool deep = !in.bgr16.empty();                   // --deep on a 16-bit PNG: CV_16UC3 through the engine (extension)
                Image src, dst;
                if (deep) {
                    out.bgr16.resize((size_t)out.rows * out.cols * 3);
                    src = Image{(uint8_t*)in.bgr16.data(), in.rows, in.cols, (size_t)in.cols * 6, 16}; dst = Image{(uint8_t*)out.bgr16.data(), out.rows, out.cols, (size_t)out.cols * 6, 16};
                } else {
                    out.bgr.resize((size_t)out.rows * out.cols * 3);
                    src = Image{in.bgr.data(), in.rows, in.cols, (size_t)in.cols * 3}; dst = Image{out.bgr.data(), out.rows, out.cols, (size_t)out.cols * 3};
                }
                auto render_still = [&](const Image& s0, Image& d0) {
                    if (o.devices == 1) return engines[0]->render(s0, d0);
                    // every tile once: engine k takes the k-th contiguous range of the tile order, the seam bands travel device to device, each
                    // engine composes and downloads its own cells of the output (Img2Img::renderSharded).  16-bit frames and --split strips keep the
                    // tile-column strips (each engine on its own host thread, the seam column recomputed instead of exchanged).
                    if (s0.depth == 8 && o.split == "shards") {
                        std::vector<Img2Img*> es;
                        for (auto& e : engines) es.push_back(e.get());
                        return Img2Img::renderSharded(es.data(), (int)es.size(), s0, d0);
                    }
                    std::vector<std::thread> th; std::vector<char> oks(o.devices, 1);
                    for (int d = 0; d < o.devices; ++d) th.emplace_back([&, d] { Image s2 = s0, d2 = d0; oks[d] = engines[d]->renderStrip(s2, d2, d, o.devices); });
                    for (auto& t : th) t.join();
                    bool all = true;
                    for (char k : oks) all = all && k;
                    return all;
                };
                const bool ok = render_still(src, dst);
                if (!ok) return -1;
                if (!in.alpha.empty()) {   // the alpha plane as a gray image through the same engine; its green channel is the new alpha
                    cli::Bitmap ga, go;
                    // sized from the geometry: with --deep the colour planes live in bgr16 and in.bgr / out.bgr are empty
                    ga.rows = in.rows; ga.cols = in.cols; ga.bgr.resize((size_t)in.rows * in.cols * 3);
                    for (size_t i = 0; i < in.alpha.size(); ++i) ga.bgr[3 * i] = ga.bgr[3 * i + 1] = ga.bgr[3 * i + 2] = in.alpha[i];
                    go.bgr.resize((size_t)out.rows * out.cols * 3);
                    Image as{ga.bgr.data(), in.rows, in.cols, (size_t)in.cols * 3}, ad{go.bgr.data(), out.rows, out.cols, (size_t)out.cols * 3};
                    if (!render_still(as, ad)) return -1;
                    out.alpha.resize((size_t)out.rows * out.cols);
                    for (size_t i = 0; i < out.alpha.size(); ++i) out.alpha[i] = go.bgr[3 * i + 1];
                }
                std::string outFile = cli::output_path(o, file, true);
                if (fs::path(file).extension() == ".bmp" || fs::path(file).extension() == ".BMP") outFile = fs::path(outFile).replace_extension(".bmp").string();   // built-in formats stay what they were, except ...
                if (fs::path(outFile).extension() == ".ppm") outFile = fs::path(outFile).replace_extension(".png").string();
                cli::write_image(outFile, out);
                ++frameIndex;
            } else {
                // a video (or a still in a format only ffmpeg reads): uncompressed AVI files through the built-in reader, the rest through ffmpeg
                std::unique_ptr<FrameSource> source; std::unique_ptr<FrameSink> sink;
                int width = 0, height = 0, frames = 1; double fps = 30.0;
                std::string why;
                auto avi = std::make_unique<AviSource>();
                const bool is_avi = fs::path(file).extension() == ".avi" || fs::path(file).extension() == ".AVI";
                if (is_avi && avi->rd.open(file, &why)) {
                    width = avi->rd.info().width; height = avi->rd.info().height; frames = avi->rd.info().frames; fps = avi->rd.info().fps;
                    source = std::move(avi);
                } else {
                    if (!have_ffmpeg) throw std::runtime_error(file + ": needs ffmpeg and ffprobe on PATH (built in: .png, .ppm, .bmp, uncompressed 24-bit .avi" + (why.empty() ? "" : "; " + why) + ")");
                    const Probe pr = ffprobe(file);
                    width = pr.width; height = pr.height; frames = pr.frames; fps = pr.fps;
                    source.reset(new PipeSource("ffmpeg -v error -i " + shell_quote(file) + " -f rawvideo -pix_fmt bgr24 -", (size_t)width * height * 3));
                }
                frameIndex = 0; frameCount = frames;
                const bool single = frames == 1;
                std::string outFile = cli::output_path(o, file, single);
                const size_t outBytes = (size_t)width * height * 3 * o.scale * o.scale;
                if (have_ffmpeg) {
                    std::string wcmd = "ffmpeg -v error -y -f rawvideo -pix_fmt bgr24 -s " + std::to_string(width * o.scale) + "x" + std::to_string(height * o.scale) +
                                       " -r " + std::to_string(single ? 1.0 : fps) + " -i - ";
                    if (!single) wcmd += "-c:v " + o.codec + " -pix_fmt " + o.pixFmt + " -crf " + std::to_string(o.crf) + " ";
                    sink.reset(new PipeSink(wcmd + shell_quote(outFile), outBytes));
                } else {   // no encoder here: the upscaled stream as an uncompressed AVI
                    outFile = fs::path(outFile).replace_extension(".avi").string();
                    on_message(Severity::info, "ffmpeg is not on PATH: writing uncompressed video to " + outFile);
                    auto as = std::make_unique<AviSink>();
                    as->wr.open(outFile, width * o.scale, height * o.scale, fps);
                    sink = std::move(as);
                }
                const bool ok = run_frame_pipeline(engines, *source, *sink, width, height, o.scale, [&](int n) { frameIndex += n; if (n) on_progress(1, 1, 0.0); });
                const bool closed = sink->close();
                if (!ok || !closed) return -1;
            }
            ++fileIndex;
        }
        return 0;
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return -1;
    }
}
