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

// Frame sizes the built-in codecs accept: headers are untrusted, and everything downstream sizes buffers as rows * cols * 3.  2^28 pixels
// (16384 x 16384) is an 805 MB frame; OpenCV, which decodes for the reference, stops at 2^30 (CV_IO_MAX_IMAGE_PIXELS).
constexpr int kMaxSide = 1 << 24;
constexpr long long kMaxPixels = 1ll << 28;
void check_dims(const std::string& path, long long w, long long h) {
    if (w <= 0 || h <= 0) throw std::runtime_error(path + ": image without a size (" + std::to_string(w) + " x " + std::to_string(h) + ")");
    if (w > kMaxSide || h > kMaxSide || w * h > kMaxPixels)
        throw std::runtime_error(path + ": " + std::to_string(w) + " x " + std::to_string(h) + " pixels is beyond what the built-in codecs take (2^28 pixels); convert through ffmpeg");
}

int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return pa <= pb && pa <= pc ? a : pb <= pc ? b : c; }

Bitmap read_png(const std::vector<uint8_t>& d, const std::string& path, bool keep16) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 33 || memcmp(d.data(), sig, 8)) throw std::runtime_error(path + ": not a PNG file");
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t pos = 8; pos + 12 <= d.size();) {
        const uint32_t len = be32(&d[pos]);
        const std::string type((const char*)&d[pos + 4], 4);
        if (pos + 12 + len > d.size()) throw std::runtime_error(path + ": truncated PNG chunk");
        const uint8_t* body = &d[pos + 8];
        if (type == "IHDR") {
            if (len < 13) throw std::runtime_error(path + ": short IHDR chunk");
            check_dims(path, be32(body), be32(body + 4));
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        }
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
    // deflate expands by at most ~1032 : 1: a header that announces more scanline bytes than the IDAT data can hold is refused before anything of
    // that size is allocated
    if (idat.empty() || total > idat.size() * 1032 + 64) throw std::runtime_error(path + ": PNG data is shorter than the image the header announces");
    std::vector<uint8_t> raw(total);
    uLongf rawlen = raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) throw std::runtime_error(path + ": PNG data does not inflate");
    const bool deep = keep16 && wide;                              // --deep: the low bytes stay
    Bitmap b; b.rows = h; b.cols = w;
    if (deep) b.bgr16.resize((size_t)w * h * 3); else b.bgr.resize((size_t)w * h * 3);
    const bool has_alpha = ctype == 4 || ctype == 6;               // kept for the writer (the network sees colour only)
    if (has_alpha) b.alpha.resize((size_t)w * h);
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
                const size_t px = (size_t)(ps.y0 + y * ps.dy) * w + ps.x0 + x * ps.dx;
                if (deep) {     // big-endian 16-bit samples
                    auto s16 = [&](int k) { const uint8_t* q = &cur[(size_t)x * bpp + (size_t)k * 2]; return (uint16_t)(q[0] << 8 | q[1]); };
                    uint16_t* o = &b.bgr16[px * 3];
                    if (ctype == 0 || ctype == 4) o[0] = o[1] = o[2] = s16(0); else { o[2] = s16(0); o[1] = s16(1); o[0] = s16(2); }
                } else {
                    uint8_t* o = &b.bgr[px * 3];
                    o[0] = bl; o[1] = g; o[2] = r;
                }
                if (has_alpha) b.alpha[px] = cur[(size_t)x * bpp + (size_t)(ch - 1) * bps];
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
    const bool rgba = !b.alpha.empty(), deep = !b.bgr16.empty();
    const int ch = rgba ? 4 : 3, bps = deep ? 2 : 1;
    std::vector<uint8_t> raw(((size_t)b.cols * ch * bps + 1) * b.rows);
    for (int y = 0; y < b.rows; ++y) {
        uint8_t* line = &raw[((size_t)b.cols * ch * bps + 1) * y];
        line[0] = 0;   // filter "none": the payload is deflated at level 1 for speed (a 4K frame is 100 MB)
        for (int x = 0; x < b.cols; ++x) {
            uint8_t* o = line + 1 + (size_t)ch * bps * x;
            if (!deep) {
                const uint8_t* s = &b.bgr[((size_t)y * b.cols + x) * 3];
                o[0] = s[2]; o[1] = s[1]; o[2] = s[0];
                if (rgba) o[3] = b.alpha[(size_t)y * b.cols + x];
            } else {        // big-endian samples; an 8-bit alpha plane is widened by x 257
                const uint16_t* s = &b.bgr16[((size_t)y * b.cols + x) * 3];
                for (int k = 0; k < 3; ++k) { o[2 * k] = (uint8_t)(s[2 - k] >> 8); o[2 * k + 1] = (uint8_t)s[2 - k]; }
                if (rgba) o[6] = o[7] = b.alpha[(size_t)y * b.cols + x];
            }
        }
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 1) != Z_OK) throw std::runtime_error("PNG deflate failed");
    comp.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a}, ihdr;
    put32(ihdr, b.cols); put32(ihdr, b.rows); ihdr.insert(ihdr.end(), {(uint8_t)(deep ? 16 : 8), (uint8_t)(rgba ? 6 : 2), 0, 0, 0});
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", comp); chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f.write((const char*)out.data(), out.size())) throw std::runtime_error("cannot write " + path);
}

// ---- device-independent bitmaps (BMP files and the frames of an uncompressed AVI): rows of BGR(A) padded to 4 bytes, bottom-up unless
// the height is negative
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | p[1] << 8); }
void putle32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x); v.push_back(x >> 8); v.push_back(x >> 16); v.push_back(x >> 24); }
void putle16(std::vector<uint8_t>& v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
size_t dib_stride(int w, int bits) { return ((size_t)w * bits + 31) / 32 * 4; }

Bitmap read_bmp(const std::vector<uint8_t>& d, const std::string& path) {
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') throw std::runtime_error(path + ": not a BMP file");
    const uint32_t off = le32(&d[10]), hdr = le32(&d[14]);
    if (hdr < 40) throw std::runtime_error(path + ": unsupported BMP header");
    const int w = (int)le32(&d[18]), hs = (int)le32(&d[22]), bits = le16(&d[28]);
    const uint32_t comp = le32(&d[30]);
    check_dims(path, w, hs < 0 ? -(long long)hs : hs);
    const int h = hs < 0 ? -hs : hs;
    if (w <= 0 || h <= 0 || (bits != 24 && bits != 32) || !(comp == 0 || (comp == 3 && bits == 32))) throw std::runtime_error(path + ": unsupported BMP (24 / 32-bit uncompressed only)");
    if (comp == 3 && (d.size() < 54 + 12 || le32(&d[54]) != 0x00FF0000u || le32(&d[58]) != 0x0000FF00u || le32(&d[62]) != 0x000000FFu)) throw std::runtime_error(path + ": unsupported BMP channel masks");
    const size_t stride = dib_stride(w, bits);
    if ((size_t)off + stride * h > d.size()) throw std::runtime_error(path + ": truncated BMP");
    Bitmap b; b.rows = h; b.cols = w; b.bgr.resize((size_t)w * h * 3);
    bool any_alpha = false;
    std::vector<uint8_t> alpha;
    if (bits == 32) alpha.resize((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* src = &d[off + stride * (size_t)(hs < 0 ? y : h - 1 - y)];
        uint8_t* o = &b.bgr[(size_t)y * w * 3];
        if (bits == 24) memcpy(o, src, (size_t)w * 3);
        else for (int x = 0; x < w; ++x) { o[3 * x] = src[4 * x]; o[3 * x + 1] = src[4 * x + 1]; o[3 * x + 2] = src[4 * x + 2]; alpha[(size_t)y * w + x] = src[4 * x + 3]; any_alpha |= src[4 * x + 3] != 0; }
    }
    if (any_alpha) b.alpha.swap(alpha);      // (32-bit files with an all-zero fourth byte carry no alpha: BI_RGB leaves it undefined)
    return b;
}

void write_bmp(const std::string& path, const Bitmap& b) {
    const bool a = !b.alpha.empty();
    const int bits = a ? 32 : 24;
    const size_t stride = dib_stride(b.cols, bits), img = stride * b.rows;
    if (54 + img > 0xFFFFFFFFull) throw std::runtime_error("image too large for a BMP file");
    std::vector<uint8_t> out;
    out.reserve(54 + img);
    out.push_back('B'); out.push_back('M'); putle32(out, (uint32_t)(54 + img)); putle32(out, 0); putle32(out, 54);
    putle32(out, 40); putle32(out, (uint32_t)b.cols); putle32(out, (uint32_t)b.rows); putle16(out, 1); putle16(out, (uint16_t)bits); putle32(out, 0);
    putle32(out, (uint32_t)img); putle32(out, 2835); putle32(out, 2835); putle32(out, 0); putle32(out, 0);
    out.resize(54 + img, 0);
    for (int y = 0; y < b.rows; ++y) {
        uint8_t* o = &out[54 + stride * (size_t)(b.rows - 1 - y)];
        const uint8_t* s = &b.bgr[(size_t)y * b.cols * 3];
        if (!a) memcpy(o, s, (size_t)b.cols * 3);
        else for (int x = 0; x < b.cols; ++x) { o[4 * x] = s[3 * x]; o[4 * x + 1] = s[3 * x + 1]; o[4 * x + 2] = s[3 * x + 2]; o[4 * x + 3] = b.alpha[(size_t)y * b.cols + x]; }
    }
    std::ofstream f(path, std::ios::binary);
    if (!f.write((const char*)out.data(), out.size())) throw std::runtime_error("cannot write " + path);
}

Bitmap read_ppm(const std::vector<uint8_t>& d, const std::string& path) {
    size_t pos = 0;
    auto token = [&]() {
        while (pos < d.size() && (isspace(d[pos]) || d[pos] == '#')) { if (d[pos] == '#') while (pos < d.size() && d[pos] != '\n') ++pos; else ++pos; }
        std::string t; while (pos < d.size() && !isspace(d[pos]) && t.size() < 16) t += (char)d[pos++];
        return t;
    };
    if (token() != "P6") throw std::runtime_error(path + ": not a binary PPM (P6)");
    auto number = [&]() { const std::string t = token(); char* end = nullptr; const long v = strtol(t.c_str(), &end, 10); if (t.empty() || *end) throw std::runtime_error(path + ": bad PPM header"); return v; };
    const long wl = number(), hl = number(), mx = number();
    check_dims(path, wl, hl);
    const int w = (int)wl, h = (int)hl;
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

bool is_builtin_still(const std::string& path) { const std::string e = lower_ext(path); return e == ".png" || e == ".ppm" || e == ".bmp"; }

Bitmap read_image(const std::string& path, bool keep16) {
    const std::vector<uint8_t> d = slurp(path);
    if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') return read_ppm(d, path);
    if (d.size() >= 2 && d[0] == 'B' && d[1] == 'M') return read_bmp(d, path);
    return read_png(d, path, keep16);
}

void write_image(const std::string& path, const Bitmap& b) {
    const size_t n = (size_t)b.rows * b.cols;
    if (!((b.bgr.size() == n * 3 && b.bgr16.empty()) || (b.bgr16.size() == n * 3 && b.bgr.empty())) || (!b.alpha.empty() && b.alpha.size() != n)) throw std::runtime_error("bitmap size mismatch");
    const std::string e = lower_ext(path);
    if (!b.bgr16.empty() && e != ".png") throw std::runtime_error(path + ": 16-bit samples can only be written as PNG");
    if (e == ".ppm") write_ppm(path, b); else if (e == ".bmp") write_bmp(path, b); else if (e == ".png") write_png(path, b);
    else throw std::runtime_error(path + ": no built-in still-image writer for this extension (.png, .ppm, .bmp)");
}

// ---- uncompressed AVI (RIFF AVI 1.0): RIFF 'AVI ' { LIST 'hdrl' { 'avih', LIST 'strl' { 'strh', 'strf' } }, LIST 'movi' { '00db' frame ... }, 'idx1' }
AviReader::~AviReader() { if (f_) fclose(f_); }

bool AviReader::open(const std::string& path, std::string* why) {
    auto no = [&](const char* w) { if (why) *why = w; if (f_) { fclose(f_); f_ = nullptr; } return false; };
    f_ = fopen(path.c_str(), "rb");
    if (!f_) return no("cannot open the file");
    uint8_t h[12];
    if (fread(h, 1, 12, f_) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "AVI ", 4)) return no("not a RIFF AVI file");
    const long riff_end = 8 + (long)le32(h + 4);
    bool have_fmt = false, video = false;
    uint32_t scale = 1, rate = 30, usec = 0;
    int bits = 0; uint32_t comp = 1; int hs = 0;
    // walk the top-level chunks up to the 'movi' list; 'hdrl' is descended into
    std::vector<long> ends = {riff_end};
    while (ftell(f_) + 8 <= riff_end) {
        uint8_t c[8];
        if (fread(c, 1, 8, f_) != 8) break;
        const uint32_t len = le32(c + 4);
        const long body = ftell(f_);
        if (!memcmp(c, "LIST", 4)) {
            uint8_t t[4];
            if (fread(t, 1, 4, f_) != 4) break;
            if (!memcmp(t, "movi", 4)) { movi_end_ = body + (long)len; break; }
            if (!memcmp(t, "hdrl", 4) || !memcmp(t, "strl", 4)) continue;      // descend
            fseek(f_, body + (long)((len + 1) & ~1u), SEEK_SET);
            continue;
        }
        std::vector<uint8_t> b(len < 4096 ? len : 4096);
        if (!b.empty() && fread(b.data(), 1, b.size(), f_) != b.size()) break;
        if (!memcmp(c, "avih", 4) && b.size() >= 40) { usec = le32(&b[0]); info_.frames = (int)le32(&b[16]); }
        else if (!memcmp(c, "strh", 4) && b.size() >= 36) { video = !memcmp(&b[0], "vids", 4) && !have_fmt; if (video) { scale = le32(&b[20]); rate = le32(&b[24]); if (le32(&b[32])) info_.frames = (int)le32(&b[32]); } }
        else if (!memcmp(c, "strf", 4) && video && !have_fmt && b.size() >= 40) { info_.width = (int)le32(&b[4]); hs = (int)le32(&b[8]); bits = le16(&b[14]); comp = le32(&b[16]); have_fmt = true; }
        fseek(f_, body + (long)((len + 1) & ~1u), SEEK_SET);
    }
    if (!have_fmt || !movi_end_) return no("no video stream / no movi list");
    if (comp != 0 || bits != 24) return no("the video stream is not uncompressed 24-bit (BI_RGB)");
    if (info_.width <= 0 || hs == 0 || hs == INT32_MIN || info_.width > kMaxSide || (hs < 0 ? -hs : hs) > kMaxSide || (long long)info_.width * (hs < 0 ? -hs : hs) > kMaxPixels) return no("bad frame size");
    info_.height = hs < 0 ? -hs : hs; bottom_up_ = hs > 0;
    info_.fps = scale && rate ? (double)rate / scale : usec ? 1e6 / usec : 30.0;
    if (info_.frames <= 0) info_.frames = 1;
    row_.resize(dib_stride(info_.width, 24));
    return true;
}

bool AviReader::read(uint8_t* bgr) {
    if (!f_) return false;
    const size_t stride = row_.size(), need = stride * info_.height;
    while (ftell(f_) + 8 <= movi_end_) {
        uint8_t c[8];
        if (fread(c, 1, 8, f_) != 8) return false;
        const uint32_t len = le32(c + 4);
        const long body = ftell(f_);
        if (!memcmp(c, "LIST", 4)) { fseek(f_, 4, SEEK_CUR); continue; }            // 'rec ' groups: descend
        const bool frame = c[0] == '0' && c[1] == '0' && c[2] == 'd' && (c[3] == 'b' || c[3] == 'c');
        if (frame && len >= need) {
            for (int y = 0; y < info_.height; ++y) {
                if (fread(row_.data(), 1, stride, f_) != stride) return false;
                memcpy(bgr + (size_t)(bottom_up_ ? info_.height - 1 - y : y) * info_.width * 3, row_.data(), (size_t)info_.width * 3);
            }
            fseek(f_, body + (long)((len + 1) & ~1u), SEEK_SET);
            return true;
        }
        fseek(f_, body + (long)((len + 1) & ~1u), SEEK_SET);                             // audio, JUNK, empty (dropped) frames
    }
    return false;
}

AviWriter::~AviWriter() { if (f_) { try { close(); } catch (...) {} } }

void AviWriter::open(const std::string& path, int width, int height, double fps) {
    f_ = fopen(path.c_str(), "wb");
    if (!f_) throw std::runtime_error("cannot write " + path);
    w_ = width; h_ = height; stride_ = dib_stride(width, 24); frames_ = 0; offsets_.clear();
    buf_.assign(stride_ * height, 0);
    // frame rate as a fraction with a millihertz grid (23.976 -> 23976 / 1000)
    scale_ = 1000; rate_ = (uint32_t)(fps * 1000.0 + 0.5); if (!rate_) { rate_ = 30000; }
    std::vector<uint8_t> hd;
    auto four = [&](const char* s) { hd.insert(hd.end(), s, s + 4); };
    four("RIFF"); putle32(hd, 0); four("AVI ");
    four("LIST"); putle32(hd, 4 + 8 + 56 + 12 + 8 + 56 + 8 + 40); four("hdrl");
    four("avih"); putle32(hd, 56);
    putle32(hd, (uint32_t)(1e6 * scale_ / rate_ + 0.5)); putle32(hd, 0); putle32(hd, 0); putle32(hd, 0x10 /* has index */); putle32(hd, 0 /* frames: patched */); putle32(hd, 0);
    putle32(hd, 1); putle32(hd, (uint32_t)buf_.size()); putle32(hd, (uint32_t)w_); putle32(hd, (uint32_t)h_); for (int i = 0; i < 4; ++i) putle32(hd, 0);
    four("LIST"); putle32(hd, 4 + 8 + 56 + 8 + 40); four("strl");
    four("strh"); putle32(hd, 56);
    four("vids"); four("DIB "); putle32(hd, 0); putle32(hd, 0); putle32(hd, 0); putle32(hd, scale_); putle32(hd, rate_); putle32(hd, 0); putle32(hd, 0 /* length: patched */);
    putle32(hd, (uint32_t)buf_.size()); putle32(hd, 0xFFFFFFFFu); putle32(hd, 0); putle16(hd, 0); putle16(hd, 0); putle16(hd, (uint16_t)w_); putle16(hd, (uint16_t)h_);
    four("strf"); putle32(hd, 40);
    putle32(hd, 40); putle32(hd, (uint32_t)w_); putle32(hd, (uint32_t)h_); putle16(hd, 1); putle16(hd, 24); putle32(hd, 0); putle32(hd, (uint32_t)buf_.size()); for (int i = 0; i < 4; ++i) putle32(hd, 0);
    four("LIST"); putle32(hd, 0); four("movi");
    movi_at_ = (long)hd.size() - 4;                 // position of the 'movi' fourcc: index offsets count from here
    if (fwrite(hd.data(), 1, hd.size(), f_) != hd.size()) throw std::runtime_error("cannot write " + path);
}

void AviWriter::write(const uint8_t* bgr) {
    if (!f_) throw std::runtime_error("AVI writer is not open");
    const long at = ftell(f_);
    if ((unsigned long long)at + 8 + buf_.size() + 16ull * (frames_ + 1) + 8 > 0xFFFFF000ull)
        throw std::runtime_error("the built-in uncompressed AVI writer ends at 4 GB per file (RIFF AVI 1.0): use ffmpeg for longer streams");
    for (int y = 0; y < h_; ++y) memcpy(&buf_[stride_ * (size_t)(h_ - 1 - y)], bgr + (size_t)y * w_ * 3, (size_t)w_ * 3);
    uint8_t c[8] = {'0', '0', 'd', 'b'};
    const uint32_t len = (uint32_t)buf_.size();
    c[4] = (uint8_t)len; c[5] = (uint8_t)(len >> 8); c[6] = (uint8_t)(len >> 16); c[7] = (uint8_t)(len >> 24);
    if (fwrite(c, 1, 8, f_) != 8 || fwrite(buf_.data(), 1, buf_.size(), f_) != buf_.size()) throw std::runtime_error("write failed (uncompressed AVI)");
    if (len & 1) fputc(0, f_);
    offsets_.push_back((uint32_t)(at - movi_at_));
    ++frames_;
}

void AviWriter::close() {
    if (!f_) return;
    FILE* f = f_; f_ = nullptr;
    const long movi_end = ftell(f);
    std::vector<uint8_t> ix;
    ix.insert(ix.end(), {'i', 'd', 'x', '1'}); putle32(ix, 16 * frames_);
    for (uint32_t k = 0; k < frames_; ++k) { ix.insert(ix.end(), {'0', '0', 'd', 'b'}); putle32(ix, 0x10); putle32(ix, offsets_[k]); putle32(ix, (uint32_t)buf_.size()); }
    bool ok = fwrite(ix.data(), 1, ix.size(), f) == ix.size();
    const long end = ftell(f);
    auto patch = [&](long pos, uint32_t v) { uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; ok = ok && !fseek(f, pos, SEEK_SET) && fwrite(b, 1, 4, f) == 4; };
    patch(4, (uint32_t)(end - 8));                                  // RIFF size
    patch(12 + 12 + 8 + 16, frames_);                               // avih.dwTotalFrames
    patch(12 + 12 + 8 + 56 + 12 + 8 + 32, frames_);                 // strh.dwLength
    patch(movi_at_ - 4, (uint32_t)(movi_end - movi_at_));           // LIST movi size
    ok = !fclose(f) && ok;
    if (!ok) throw std::runtime_error("cannot finish the AVI file");
}

}  // namespace w2x::cli
