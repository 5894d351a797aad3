// Flat C ABI over w2x::Img2Img (declared in include/w2x/c_api.h, which cites the reference interfaces).
#include "../../include/w2x/c_api.h"

#include <cstring>
#include <fstream>
#include <vector>
#include <string>

#include "../../include/w2x/img2img.h"
#include "lower.h"
#include "sha256.h"
#include "tiles.h"
#include "kernels.h"

struct w2x_engine {
    w2x::Img2Img engine;
    w2x_message_fn msg = nullptr; void* msg_user = nullptr;
    w2x_progress_fn prog = nullptr; void* prog_user = nullptr;
};

extern "C" {

w2x_engine* w2x_create(void) { try { return new w2x_engine; } catch (...) { return nullptr; } }
void w2x_destroy(w2x_engine* e) { delete e; }

void w2x_set_message_callback(w2x_engine* e, w2x_message_fn fn, void* user) {
    if (!e) return;
    e->msg = fn; e->msg_user = user;
    if (fn) e->engine.setMessageCallback([e](w2x::Severity s, const std::string& m) { if (e->msg) e->msg((int)s, m.c_str(), e->msg_user); });
    else e->engine.setMessageCallback(nullptr);
}
void w2x_set_progress_callback(w2x_engine* e, w2x_progress_fn fn, void* user) {
    if (!e) return;
    e->prog = fn; e->prog_user = user;
    if (fn) e->engine.setProgressCallback([e](int c, int t, double s) { if (e->prog) e->prog(c, t, s, e->prog_user); });
    else e->engine.setProgressCallback(nullptr);
}

int w2x_build(w2x_engine* e, const char* onnx_path, const w2x_build_config* c) {
    if (!e || !onnx_path || !c) return 0;
    w2x::BuildConfig b;
    b.deviceId = c->deviceId; b.precision = c->precision == W2X_PRECISION_FP16 ? w2x::Precision::FP16 : c->precision == W2X_PRECISION_FP32 ? w2x::Precision::FP32 : w2x::Precision::TF32;
    b.minBatchSize = c->minBatchSize; b.optBatchSize = c->optBatchSize; b.maxBatchSize = c->maxBatchSize;
    b.minChannels = c->minChannels; b.optChannels = c->optChannels; b.maxChannels = c->maxChannels;
    b.minWidth = c->minWidth; b.optWidth = c->optWidth; b.maxWidth = c->maxWidth;
    b.minHeight = c->minHeight; b.optHeight = c->optHeight; b.maxHeight = c->maxHeight;
    return e->engine.build(onnx_path, b) ? 1 : 0;
}

int w2x_load(w2x_engine* e, const char* onnx_path, const w2x_render_config* c) {
    if (!e || !onnx_path || !c) return 0;
    w2x::RenderConfig r;
    r.deviceId = c->deviceId; r.precision = c->precision == W2X_PRECISION_FP16 ? w2x::Precision::FP16 : c->precision == W2X_PRECISION_FP32 ? w2x::Precision::FP32 : w2x::Precision::TF32;
    r.batchSize = c->batchSize; r.channels = c->channels; r.height = c->height; r.width = c->width; r.scaling = c->scaling;
    r.overlapX = c->overlapX; r.overlapY = c->overlapY; r.tta = c->tta != 0; r.ttaBugCompat = c->ttaBugCompat != 0;
    return e->engine.load(onnx_path, r) ? 1 : 0;
}

int w2x_render(w2x_engine* e, const uint8_t* src, int rows, int cols, size_t src_step, uint8_t* dst, size_t dst_step) {
    if (!e) return 0;
    w2x::Image s; s.data = const_cast<uint8_t*>(src); s.rows = rows; s.cols = cols; s.step = src_step;
    // the caller pre-sizes dst to size*scale (main.cpp:234-235); the scale is the engine's
    w2x::Image d; d.data = dst; d.step = dst_step;
    const int sc = e->engine.scaling();
    d.rows = rows * sc; d.cols = cols * sc;
    return e->engine.render(s, d) ? 1 : 0;
}

int w2x_render16(w2x_engine* e, const uint16_t* src, int rows, int cols, size_t src_step, uint16_t* dst, size_t dst_step) {
    if (!e) return 0;
    w2x::Image s; s.data = reinterpret_cast<uint8_t*>(const_cast<uint16_t*>(src)); s.rows = rows; s.cols = cols; s.step = src_step; s.depth = 16;
    w2x::Image d; d.data = reinterpret_cast<uint8_t*>(dst); d.step = dst_step; d.depth = 16;
    const int sc = e->engine.scaling();
    d.rows = rows * sc; d.cols = cols * sc;
    return e->engine.render(s, d) ? 1 : 0;
}

int w2x_render_strip(w2x_engine* e, const uint8_t* src, int rows, int cols, size_t src_step, uint8_t* dst, size_t dst_step, int part, int parts) {
    if (!e) return 0;
    w2x::Image s; s.data = const_cast<uint8_t*>(src); s.rows = rows; s.cols = cols; s.step = src_step;
    w2x::Image d; d.data = dst; d.step = dst_step;
    const int sc = e->engine.scaling();
    d.rows = rows * sc; d.cols = cols * sc;
    return e->engine.renderStrip(s, d, part, parts) ? 1 : 0;
}

int w2x_render_sharded(w2x_engine* const* engines, int count, const uint8_t* src, int rows, int cols, size_t src_step, uint8_t* dst, size_t dst_step) {
    if (!engines || count <= 0) return 0;
    std::vector<w2x::Img2Img*> es(count);
    for (int k = 0; k < count; ++k) { if (!engines[k]) return 0; es[k] = &engines[k]->engine; }
    w2x::Image s; s.data = const_cast<uint8_t*>(src); s.rows = rows; s.cols = cols; s.step = src_step;
    w2x::Image d; d.data = dst; d.step = dst_step;
    const int sc = es[0]->scaling();
    d.rows = rows * sc; d.cols = cols * sc;
    return w2x::Img2Img::renderSharded(es.data(), count, s, d) ? 1 : 0;
}

int w2x_shard_compute(w2x_engine* e, const uint8_t* src, int rows, int cols, size_t src_step, int part, int parts) {
    if (!e) return 0;
    w2x::Image s; s.data = const_cast<uint8_t*>(src); s.rows = rows; s.cols = cols; s.step = src_step;
    return e->engine.shardCompute(s, part, parts) ? 1 : 0;
}
const void* w2x_shard_slab(w2x_engine* e, size_t* bytes) { return e ? e->engine.shardSlab(bytes) : nullptr; }
int w2x_shard_finish(w2x_engine* e, uint8_t* dst, int rows, int cols, size_t dst_step, int part, int parts, const void* const* slabs, const int* devices) {
    if (!e) return 0;
    w2x::Image d; d.data = dst; d.rows = rows; d.cols = cols; d.step = dst_step;
    return e->engine.shardFinish(d, part, parts, slabs, devices) ? 1 : 0;
}
int w2x_ipc_export(const void* device_ptr, uint8_t* out64) { return out64 && w2x::ipc_export(device_ptr, out64) ? 1 : 0; }
void* w2x_ipc_open(const uint8_t* handle64, int device) { return handle64 ? w2x::ipc_open(handle64, device) : nullptr; }
void w2x_ipc_close(void* p) { w2x::ipc_close(p); }

int w2x_render_sequence(w2x_engine* e, const uint8_t* const* srcs, int rows, int cols, size_t src_step, uint8_t* const* dsts, size_t dst_step, int count) {
    if (!e || count < 0 || (count > 0 && (!srcs || !dsts))) return 0;
    const int sc = e->engine.scaling();
    std::vector<w2x::Image> s(count), d(count);
    for (int i = 0; i < count; ++i) {
        s[i].data = const_cast<uint8_t*>(srcs[i]); s[i].rows = rows; s[i].cols = cols; s[i].step = src_step;
        d[i].data = dsts[i]; d[i].rows = rows * sc; d[i].cols = cols * sc; d[i].step = dst_step;
    }
    return e->engine.renderSequence(s.data(), d.data(), count) ? 1 : 0;
}
void* w2x_alloc_host(w2x_engine* e, size_t bytes) { return e ? e->engine.allocHost(bytes) : nullptr; }
void w2x_free_host(w2x_engine* e, void* data) { if (e) e->engine.freeHost(data); }
int w2x_pin_host(w2x_engine* e, void* data, size_t bytes) { return e && e->engine.pinHost(data, bytes) ? 1 : 0; }
void w2x_unpin_host(w2x_engine* e, void* data) { if (e) e->engine.unpinHost(data); }

int w2x_strip_plan(int in_w, int in_h, int out_w, int out_h, int tile_in, int tile_out, int scaling, double overlap_x, double overlap_y,
                   int part, int parts, int* out4) {
    if (!out4) return 0;
    w2x::TileGrid g = w2x::calculate_tiles(in_w, in_h, out_w, out_h, tile_in, tile_in, tile_out, tile_out, scaling, overlap_x, overlap_y);
    w2x::StripPlan sp = w2x::strip_plan(g, out_w, tile_out, part, parts);
    out4[0] = sp.first_tile; out4[1] = sp.tile_count; out4[2] = sp.x0; out4[3] = sp.x1;
    return 1;
}

int w2x_shard_plan(int in_w, int in_h, int out_w, int out_h, int tile_in, int tile_out, int scaling, double overlap_x, double overlap_y,
                   int part, int parts, int* out16) {
    if (!out16) return 0;
    w2x::TileGrid g = w2x::calculate_tiles(in_w, in_h, out_w, out_h, tile_in, tile_in, tile_out, tile_out, scaling, overlap_x, overlap_y);
    w2x::ShardPlan sp = w2x::shard_plan(g, out_w, out_h, tile_out, tile_out, part, parts);
    out16[0] = sp.first_tile; out16[1] = sp.tile_count; out16[2] = sp.halo_first; out16[3] = sp.nrect;
    for (int r = 0; r < 3; ++r) { out16[4 + 4 * r] = sp.rect[r].x; out16[5 + 4 * r] = sp.rect[r].y; out16[6 + 4 * r] = sp.rect[r].w; out16[7 + 4 * r] = sp.rect[r].h; }
    return 1;
}

int w2x_infer(w2x_engine* e, const float* in, float* out) { return e && e->engine.infer(in, out) ? 1 : 0; }
int w2x_output_tile_size(w2x_engine* e) { return e ? e->engine.outputTileSize() : 0; }
int w2x_pass_tiles(w2x_engine* e) { return e ? e->engine.passTiles() : 0; }
double w2x_plan_flops(w2x_engine* e) { return e ? e->engine.planFlops() : 0.0; }
float w2x_last_render_ms(w2x_engine* e) { return e ? e->engine.lastRenderMs() : -1.f; }
int w2x_profile_frame(w2x_engine* e, double* out, int cap) { return e && e->engine.profileFrame(out, cap) ? 1 : 0; }
int w2x_op_times(w2x_engine* e, double* out, int cap) { return e ? e->engine.opTimes(out, cap) : 0; }
float w2x_bench_resident(w2x_engine* e, int iters) { return e ? e->engine.benchResident(iters) : -1.f; }

int w2x_calculate_tiles(int in_w, int in_h, int out_w, int out_h, int tile_in, int tile_out, int scaling,
                        double overlap_x, double overlap_y, int* in_rects, int* out_rects, int cap) {
    w2x::TileGrid g = w2x::calculate_tiles(in_w, in_h, out_w, out_h, tile_in, tile_in, tile_out, tile_out, scaling, overlap_x, overlap_y);
    if (g.count > cap) return -1;
    for (int i = 0; i < g.count; ++i) {
        if (in_rects) { in_rects[4 * i] = g.in[i].x; in_rects[4 * i + 1] = g.in[i].y; in_rects[4 * i + 2] = g.in[i].w; in_rects[4 * i + 3] = g.in[i].h; }
        if (out_rects) { out_rects[4 * i] = g.out[i].x; out_rects[4 * i + 1] = g.out[i].y; out_rects[4 * i + 2] = g.out[i].w; out_rects[4 * i + 3] = g.out[i].h; }
    }
    return g.count;
}

int w2x_tile_weights(int which, int overlap_x, int overlap_y, int size, float* out) {
    if (which < 0 || which > 3 || size <= 0 || !out) return 0;
    auto m = w2x::tile_weight_mask(which, overlap_x, overlap_y, size);
    memcpy(out, m.data(), m.size() * sizeof(float));
    return 1;
}

int w2x_describe_plan(const char* onnx_path, int batch, int tile, char* buf, size_t cap) { return w2x_describe_plan_precision(onnx_path, batch, tile, W2X_PRECISION_FP16, buf, cap); }

int w2x_describe_plan_precision(const char* onnx_path, int batch, int tile, int precision, char* buf, size_t cap) {
    std::string s; int ok = 1;
    try {
        w2x::Plan plan = w2x::build_plan(onnx_path, batch, 3, tile, tile, precision != W2X_PRECISION_FP16);
        plan.userB = batch;
        const auto bytes = plan.serialize();                          // what build() writes must be what load() accepts
        s = w2x::Plan::deserialize(bytes.data(), bytes.size()).describe();
    }
    catch (const std::exception& e) { s = std::string("ERROR: ") + e.what(); ok = 0; }
    if (buf && cap) { size_t n = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(buf, s.data(), n); buf[n] = 0; }
    return ok;
}

int w2x_validate_engine_file(const char* path, char* buf, size_t cap) {
    std::string s = "ok"; int ok = 1;
    try {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (!f.is_open()) throw std::runtime_error("could not open engine file");
        std::vector<char> bytes((size_t)f.tellg());
        f.seekg(0); f.read(bytes.data(), (std::streamsize)bytes.size());
        (void)w2x::Plan::deserialize((const uint8_t*)bytes.data(), bytes.size());
    } catch (const std::exception& e) { s = e.what(); ok = 0; }
    if (buf && cap) { size_t n = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(buf, s.data(), n); buf[n] = 0; }
    return ok;
}

int w2x_write_engine_file(const char* onnx_path, int batch, int tile, const char* out_path) {
    try {
        w2x::Plan plan = w2x::build_plan(onnx_path, batch, 3, tile, tile);
        plan.userB = batch;
        const auto bytes = plan.serialize();
        std::ofstream f(out_path, std::ios::binary);
        if (!f.is_open()) return 0;
        f.write((const char*)bytes.data(), (std::streamsize)bytes.size());
        return f.good() ? 1 : 0;
    } catch (const std::exception&) { return 0; }
}

int w2x_device_pci_bus_id(int device, char* buf, size_t cap) { return buf && cap >= 16 && w2x::device_pci_bus_id(device, buf, cap) ? 1 : 0; }

void w2x_sha256_hex(const void* data, size_t len, char* out) {
    std::string h = w2x::sha256_hex(data, len);
    memcpy(out, h.c_str(), 65);
}

const char* w2x_version(void) { return "w2x-hip 0.1 (gfx950)"; }

int w2x_debug_set(const char* name, int value) { return w2x::set_switch(name, value) ? 1 : 0; }

}  // extern "C"
