#include "imageio.h"

#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace w2x::cli {

namespace {

std::string lower_ext(const std::string& p) {
    auto dot = p.rfind('.');
    std::string e = dot == std::string::npos ? "" : p.substr(dot);
    std::transform(e.begin(), e.end(), e.begin(), ::tolower);
    return e;
}

std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return pa <= pb && pa <= pc ? a : pb <= pc ? b : c; }

Bitmap read_png(const std::vector<uint8_t>& d, const std::string& path) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 33 || memcmp(d.data(), sig, 8)) throw std::runtime_error(path + ": not a PNG file");
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t pos = 8; pos + 12 <= d.size();) {
        const uint32_t len = be32(&d[pos]);
        const std::string type((const char*)&d[pos + 4], 4);
        if (pos + 12 + len > d.size()) throw std::runtime_error(path + ": truncated PNG chunk");
        const uint8_t* body = &d[pos + 8];
        if (type == "IHDR") { w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (type == "PLTE") plte.assign(body, body + len);
        else if (type == "IDAT") idat.insert(idat.end(), body, body + len);
        else if (type == "IEND") break;
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0) throw std::runtime_error(path + ": PNG without IHDR");
    const bool packed = depth < 8 && (ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4);   // sub-byte gray / palette
    // 16-bit samples keep their high byte, what cv::imread(IMREAD_COLOR) hands the reference (libpng strip_16); alpha is dropped;
    // Adam7-interlaced files are seven sub-images, each filtered on its own
    const bool wide = depth == 16 && ctype != 3;
    if ((depth != 8 && !packed && !wide) || interlace > 1) throw std::runtime_error(path + ": unsupported PNG bit depth / interlace method");
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) throw std::runtime_error(path + ": unsupported PNG colour type");
    const int bps = wide ? 2 : 1, bpp = ch * bps;                    // bytes per sample / per pixel (the filters' left neighbour)
    struct Pass { int x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const int npass = interlace ? 7 : 1;
    auto pass_dims = [&](const Pass& ps, int& pw, int& ph, size_t& stride) {
        pw = (w - ps.x0 + ps.dx - 1) / ps.dx; ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw < 0) pw = 0;
        if (ph < 0) ph = 0;
        stride = packed ? ((size_t)pw * depth + 7) / 8 : (size_t)pw * bpp;
    };
    size_t total = 0;
    for (int k = 0; k < npass; ++k) { int pw, ph; size_t st; pass_dims(interlace ? adam7[k] : whole, pw, ph, st); if (pw && ph) total += (st + 1) * ph; }
    std::vector<uint8_t> raw(total);
    uLongf rawlen = raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) throw std::runtime_error(path + ": PNG data does not inflate");
    Bitmap b; b.rows = h; b.cols = w; b.bgr.resize((size_t)w * h * 3);
    size_t at = 0;
    for (int k = 0; k < npass; ++k) {
        const Pass& ps = interlace ? adam7[k] : whole;
        int pw, ph; size_t stride;
        pass_dims(ps, pw, ph, stride);
        if (!pw || !ph) continue;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (int y = 0; y < ph; ++y) {
            const uint8_t* line = &raw[at]; at += stride + 1;
            const int ft = line[0];
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, up = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0, x = line[1 + i];
                int v;
                switch (ft) { case 0: v = x; break; case 1: v = x + a; break; case 2: v = x + up; break; case 3: v = x + ((a + up) >> 1); break; case 4: v = x + paeth(a, up, c); break;
                              default: throw std::runtime_error(path + ": bad PNG filter"); }
                cur[i] = (uint8_t)v;
            }
            for (int x = 0; x < pw; ++x) {
                uint8_t r, g, bl;
                uint8_t sample = 0;
                if (packed) { const int per = 8 / depth, sh = (per - 1 - x % per) * depth; sample = (cur[x / per] >> sh) & ((1 << depth) - 1); }
                if (ctype == 0 && packed) r = g = bl = (uint8_t)(sample * 255 / ((1 << depth) - 1));
                else if (ctype == 0 || ctype == 4) r = g = bl = cur[(size_t)x * bpp];
                else if (ctype == 3) { const size_t kk = (size_t)(packed ? sample : cur[x]) * 3; if (kk + 3 > plte.size()) throw std::runtime_error(path + ": palette index out of range"); r = plte[kk]; g = plte[kk + 1]; bl = plte[kk + 2]; }
                else { r = cur[(size_t)x * bpp]; g = cur[(size_t)x * bpp + bps]; bl = cur[(size_t)x * bpp + 2 * bps]; }
                uint8_t* o = &b.bgr[((size_t)(ps.y0 + y * ps.dy) * w + ps.x0 + x * ps.dx) * 3];
                o[0] = bl; o[1] = g; o[2] = r;
            }
            prev.swap(cur);
        }
    }
    return b;
}

void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body) {
    put32(out, (uint32_t)body.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put32(out, (uint32_t)crc32(0, &out[at], (uInt)(out.size() - at)));
}

void write_png(const std::string& path, const Bitmap& b) {
    std::vector<uint8_t> raw(((size_t)b.cols * 3 + 1) * b.rows);
    for (int y = 0; y < b.rows; ++y) {
        uint8_t* line = &raw[((size_t)b.cols * 3 + 1) * y];
        line[0] = 0;   // filter "none": the payload is deflated at level 1 for speed (a 4K frame is 100 MB)
        const uint8_t* s = &b.bgr[(size_t)y * b.cols * 3];
        for (int x = 0; x < b.cols; ++x) { line[1 + 3 * x] = s[3 * x + 2]; line[2 + 3 * x] = s[3 * x + 1]; line[3 + 3 * x] = s[3 * x]; }
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 1) != Z_OK) throw std::runtime_error("PNG deflate failed");
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a}, ihdr;
    put32(ihdr, b.cols); put32(ihdr, b.rows); ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", comp); chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f.write((const char*)out.data(), out.size())) throw std::runtime_error("cannot write " + path);
}

Bitmap read_ppm(const std::vector<uint8_t>& d, const std::string& path) {
    size_t pos = 0;
    auto token = [&]() {
        while (pos < d.size() && (isspace(d[pos]) || d[pos] == '#')) { if (d[pos] == '#') while (pos < d.size() && d[pos] != '\n') ++pos; else ++pos; }
        std::string t; while (pos < d.size() && !isspace(d[pos])) t += (char)d[pos++];
        return t;
    };
    if (token() != "P6") throw std::runtime_error(path + ": not a binary PPM (P6)");
    const int w = atoi(token().c_str()), h = atoi(token().c_str()), mx = atoi(token().c_str());
    ++pos;
    if (w <= 0 || h <= 0 || mx != 255 || pos + (size_t)w * h * 3 > d.size()) throw std::runtime_error(path + ": unsupported or truncated PPM");
    Bitmap b; b.rows = h; b.cols = w; b.bgr.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; ++i) { b.bgr[3 * i] = d[pos + 3 * i + 2]; b.bgr[3 * i + 1] = d[pos + 3 * i + 1]; b.bgr[3 * i + 2] = d[pos + 3 * i]; }
    return b;
}

void write_ppm(const std::string& path, const Bitmap& b) {
    std::ofstream f(path, std::ios::binary);
    f << "P6\n" << b.cols << " " << b.rows << "\n255\n";
    std::vector<uint8_t> rgb(b.bgr.size());
    for (size_t i = 0; i < b.bgr.size() / 3; ++i) { rgb[3 * i] = b.bgr[3 * i + 2]; rgb[3 * i + 1] = b.bgr[3 * i + 1]; rgb[3 * i + 2] = b.bgr[3 * i]; }
    if (!f.write((const char*)rgb.data(), rgb.size())) throw std::runtime_error("cannot write " + path);
}

}  // namespace

bool is_builtin_still(const std::string& path) { const std::string e = lower_ext(path); return e == ".png" || e == ".ppm"; }

Bitmap read_image(const std::string& path) {
    const std::vector<uint8_t> d = slurp(path);
    if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') return read_ppm(d, path);
    return read_png(d, path);
}

void write_image(const std::string& path, const Bitmap& b) {
    if ((size_t)b.rows * b.cols * 3 != b.bgr.size()) throw std::runtime_error("bitmap size mismatch");
    if (lower_ext(path) == ".ppm") write_ppm(path, b); else write_png(path, b);
}

}  // namespace w2x::cli
