// Still-image and raw-video I/O for the CLI: 8-bit BGR interleaved buffers (what the reference pipes through ffmpeg as bgr24,
// videoio/capture.cpp:96-99, writer.cpp:24-33).  Built-in codecs:
//   PNG (zlib; 1..16-bit gray / RGB / RGBA / palette, Adam7 or not; 16-bit samples keep their high byte like cv::imread(IMREAD_COLOR);
//        the alpha channel is kept in Bitmap::alpha and written back as RGBA - upstream lists alpha as a TODO, README.md:88),
//   binary PPM (P6), BMP (24 / 32-bit uncompressed, either row order),
//   uncompressed AVI (one video stream of 24-bit BI_RGB frames = ffmpeg's `-c:v rawvideo -pix_fmt bgr24` in an .avi): the frame loop of
//        main.cpp:263-269 runs without ffmpeg on such files; RIFF AVI 1.0, so a file ends below 4 GB.
// Everything else goes through ffmpeg.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace w2x::cli {

struct Bitmap {
    int rows = 0, cols = 0;
    std::vector<uint8_t> bgr;       // rows * cols * 3, packed (empty when bgr16 is used)
    std::vector<uint8_t> alpha;     // rows * cols, or empty
    std::vector<uint16_t> bgr16;    // 16-bit samples of a 16-bit PNG read with keep16 (--deep); then bgr is empty
};

Bitmap read_image(const std::string& path, bool keep16 = false);   // throws std::runtime_error; keep16: 16-bit PNG samples stay 16-bit (Bitmap::bgr16)
void write_image(const std::string& path, const Bitmap& b);  // by extension: .png, .ppm, .bmp
bool is_builtin_still(const std::string& path);              // .png / .ppm / .bmp

// ---- uncompressed AVI, streamed frame by frame (frames are handed over top-down, packed bgr24, like the ffmpeg pipes)
struct AviInfo { int width = 0, height = 0, frames = 0; double fps = 0; };
class AviReader {
public:
    ~AviReader();
    // false (and nothing opened) when the file is not an AVI this reader takes (no RIFF AVI header, compressed or non-24-bit video stream)
    bool open(const std::string& path, std::string* why = nullptr);
    const AviInfo& info() const { return info_; }
    bool read(uint8_t* bgr);                                  // next frame; false at the end of the stream
private:
    FILE* f_ = nullptr; AviInfo info_; long movi_end_ = 0; bool bottom_up_ = true; std::vector<uint8_t> row_;
};
class AviWriter {
public:
    ~AviWriter();
    void open(const std::string& path, int width, int height, double fps);   // throws
    void write(const uint8_t* bgr);                                          // throws (also when the file would pass 4 GB)
    void close();                                                            // patches the headers, writes the index
private:
    FILE* f_ = nullptr; int w_ = 0, h_ = 0; size_t stride_ = 0; uint32_t frames_ = 0; long movi_at_ = 0; std::vector<uint8_t> buf_; std::vector<uint32_t> offsets_;
    uint32_t rate_ = 30, scale_ = 1;
};

}  // namespace w2x::cli
