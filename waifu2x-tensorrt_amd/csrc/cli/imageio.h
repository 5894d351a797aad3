// Still-image I/O for the CLI: 8-bit BGR interleaved buffers (what the reference pipes through ffmpeg as bgr24,
// videoio/capture.cpp:96-99, writer.cpp:24-33).  Built-in codecs: PNG (zlib; 8-bit gray / RGB / RGBA / palette, non-interlaced;
// alpha is dropped - the reference does the same, README.md:88) and binary PPM (P6).  Everything else goes through ffmpeg.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace w2x::cli {

struct Bitmap { int rows = 0, cols = 0; std::vector<uint8_t> bgr; };   // rows * cols * 3, packed

Bitmap read_image(const std::string& path);                 // throws std::runtime_error
void write_image(const std::string& path, const Bitmap& b);  // by extension: .png, .ppm
bool is_builtin_still(const std::string& path);              // .png / .ppm

}  // namespace w2x::cli
