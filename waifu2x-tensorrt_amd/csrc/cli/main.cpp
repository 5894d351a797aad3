// w2x: the reference's command line (src/main.cpp) on top of w2x::Img2Img.
//   w2x --model swin_unet/art --scale 4 --noise 3 --batchSize 4 --tileSize 256 build
//   w2x --model swin_unet/art --scale 4 --noise 3 --batchSize 4 --tileSize 256 render -i in.png -o outdir [--tta] [--blend 1/16]
// Stills (.png/.ppm) are read and written by the built-in codecs; other formats and videos are piped through ffmpeg as
// raw bgr24 when ffmpeg/ffprobe are on PATH (videoio/capture.cpp:96-99, writer.cpp:24-33 do the same).
// Extension: --devices N drives N engines from N host threads - video frames go round-robin, a single image is split into
// tile-column strips (Img2Img::renderStrip), every engine writing its own columns of the shared output buffer.
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <condition_variable>
#include <csignal>
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
        size_t fileIndex = 0, fileCount = 0, frameIndex = 0, frameCount = 0;
        auto on_message = [](Severity s, const std::string& m) {
            static const char* names[] = {"critical", "error", "warning", "info", "debug", "trace"};
            std::cerr << "[" << names[(int)s < 6 ? (int)s : 5] << "] " << m << "\n";
        };
        auto on_progress = [&](int current, int total, double speed) {
            fprintf(stderr, "[info] Rendered file %zu/%zu, frame %zu/%zu, batch %d/%d @ %.2f it/s\n", fileIndex, fileCount, frameIndex, frameCount, current, total, speed);
        };
        const Precision prec = o.precision == "tf32" ? Precision::TF32 : Precision::FP16;

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
            c.overlapX = c.overlapY = o.blend; c.tta = o.tta;
            if (!engines.back()->load(modelPath, c)) return -1;
        }
        const std::vector<std::string> files = find_inputs(o);
        fileCount = files.size();
        for (const std::string& file : files) {
            if (cli::is_builtin_still(file)) {
                frameIndex = 0; frameCount = 1;
                cli::Bitmap in = cli::read_image(file), out;
                out.rows = in.rows * o.scale; out.cols = in.cols * o.scale; out.bgr.resize((size_t)out.rows * out.cols * 3);
                Image src{in.bgr.data(), in.rows, in.cols, (size_t)in.cols * 3}, dst{out.bgr.data(), out.rows, out.cols, (size_t)out.cols * 3};
                bool ok = true;
                if (o.devices == 1) ok = engines[0]->render(src, dst);
                else {   // tile-column strips: each engine composes and downloads its own columns of `out`
                    std::vector<std::thread> th; std::vector<char> oks(o.devices, 1);
                    for (int d = 0; d < o.devices; ++d) th.emplace_back([&, d] { Image s2 = src, d2 = dst; oks[d] = engines[d]->renderStrip(s2, d2, d, o.devices); });
                    for (auto& t : th) t.join();
                    for (char k : oks) ok = ok && k;
                }
                if (!ok) return -1;
                cli::write_image(cli::output_path(o, file, true), out);
                ++frameIndex;
            } else {
                if (!have_ffmpeg) throw std::runtime_error(file + ": needs ffmpeg and ffprobe on PATH (built in: .png, .ppm)");
                const Probe pr = ffprobe(file);
                frameIndex = 0; frameCount = pr.frames;
                const bool single = pr.frames == 1;
                const std::string outFile = cli::output_path(o, file, single);
                const size_t inBytes = (size_t)pr.width * pr.height * 3, outBytes = inBytes * o.scale * o.scale;
                FILE* rd = popen(("ffmpeg -v error -i " + shell_quote(file) + " -f rawvideo -pix_fmt bgr24 -").c_str(), "r");
                std::string wcmd = "ffmpeg -v error -y -f rawvideo -pix_fmt bgr24 -s " + std::to_string(pr.width * o.scale) + "x" + std::to_string(pr.height * o.scale) +
                                   " -r " + std::to_string(single ? 1.0 : pr.fps) + " -i - ";
                if (!single) wcmd += "-c:v " + o.codec + " -pix_fmt " + o.pixFmt + " -crf " + std::to_string(o.crf) + " ";
                FILE* wr = popen((wcmd + shell_quote(outFile)).c_str(), "w");
                if (!rd || !wr) { if (rd) pclose(rd); if (wr) pclose(wr); throw std::runtime_error("cannot start ffmpeg for " + file); }
                if (o.devices == 1) {
                    // one device: chunks of frames through renderSequence (upload / compute / download overlapped, buffers
                    // page-locked once).  The ffmpeg pipes run on their own threads over two chunk slots, so decoding chunk
                    // n+1 and encoding chunk n-1 overlap with rendering chunk n (the reference serialises them, main.cpp:263-269).
                    const int CH = 4, SLOTS = 2;
                    // frame buffers from the engine's page-locked allocator: their copies run by DMA beside the kernels
                    struct Slot { std::vector<uint8_t*> in, out; int frames = 0; int state = 0; };   // 0 free, 1 read, 2 rendered
                    std::vector<Slot> slots(SLOTS);
                    for (Slot& sl : slots)
                        for (int k = 0; k < CH; ++k) {
                            sl.in.push_back((uint8_t*)engines[0]->allocHost(inBytes)); sl.out.push_back((uint8_t*)engines[0]->allocHost(outBytes));
                            if (!sl.in.back() || !sl.out.back()) throw std::runtime_error("cannot allocate page-locked frame buffers");
                        }
                    std::mutex mu; std::condition_variable cv;
                    bool failed = false;
                    std::thread reader([&] {          // slot i: free -> read; a slot with 0 frames marks the end of the stream
                        for (int i = 0;; i = (i + 1) % SLOTS) {
                            Slot& sl = slots[i];
                            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sl.state == 0 || failed; }); if (failed) return; }
                            int got = 0;
                            for (; got < CH; ++got) if (fread(sl.in[got], 1, inBytes, rd) != inBytes) break;
                            { std::lock_guard<std::mutex> lk(mu); sl.frames = got; sl.state = 1; }
                            cv.notify_all();
                            if (got < CH) return;
                        }
                    });
                    std::thread writer([&] {          // slot i: rendered -> free
                        for (int i = 0;; i = (i + 1) % SLOTS) {
                            Slot& sl = slots[i];
                            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sl.state == 2 || failed; }); if (failed) return; }
                            const int n = sl.frames;
                            for (int k = 0; k < n; ++k) if (fwrite(sl.out[k], 1, outBytes, wr) != outBytes) { std::lock_guard<std::mutex> lk(mu); failed = true; }
                            { std::lock_guard<std::mutex> lk(mu); sl.state = 0; }
                            cv.notify_all();
                            if (n < CH) return;
                        }
                    });
                    for (int i = 0;; i = (i + 1) % SLOTS) {
                        Slot& sl = slots[i];
                        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sl.state == 1 || failed; }); if (failed) break; }
                        const int got = sl.frames;
                        std::vector<Image> si(got), di(got);
                        for (int k = 0; k < got; ++k) {
                            si[k] = Image{sl.in[k], pr.height, pr.width, (size_t)pr.width * 3};
                            di[k] = Image{sl.out[k], pr.height * o.scale, pr.width * o.scale, (size_t)pr.width * o.scale * 3};
                        }
                        const bool ok = got == 0 || engines[0]->renderSequence(si.data(), di.data(), got);
                        { std::lock_guard<std::mutex> lk(mu); if (!ok) failed = true; sl.state = 2; frameIndex += got; }
                        cv.notify_all();
                        if (got) on_progress(1, 1, 0.0);
                        if (!ok || got < CH) break;
                    }
                    reader.join(); writer.join();
                    for (Slot& sl : slots) for (int k = 0; k < CH; ++k) { engines[0]->freeHost(sl.in[k]); engines[0]->freeHost(sl.out[k]); }
                    pclose(rd); pclose(wr);
                    if (failed) return -1;
                    ++fileIndex;
                    continue;
                }
                // frames round-robin over the devices, written in order
                const int N = o.devices;
                std::vector<std::vector<uint8_t>> ins(N, std::vector<uint8_t>(inBytes)), outs(N, std::vector<uint8_t>(outBytes));
                bool eof = false;
                while (!eof) {
                    int got = 0;
                    for (; got < N; ++got) if (fread(ins[got].data(), 1, inBytes, rd) != inBytes) { eof = true; break; }
                    std::vector<std::thread> th; std::vector<char> oks(N, 1);
                    for (int d = 0; d < got; ++d) th.emplace_back([&, d] {
                        Image s2{ins[d].data(), pr.height, pr.width, (size_t)pr.width * 3}, d2{outs[d].data(), pr.height * o.scale, pr.width * o.scale, (size_t)pr.width * o.scale * 3};
                        oks[d] = engines[d]->render(s2, d2);
                    });
                    for (auto& t : th) t.join();
                    bool bad = false;
                    for (int d = 0; d < got && !bad; ++d) { bad = !oks[d] || fwrite(outs[d].data(), 1, outBytes, wr) != outBytes; ++frameIndex; }
                    if (bad) { pclose(rd); pclose(wr); return -1; }
                }
                pclose(rd); pclose(wr);
            }
            ++fileIndex;
        }
        return 0;
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return -1;
    }
}
