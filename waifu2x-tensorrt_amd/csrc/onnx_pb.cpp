#include "onnx_pb.h"

#include <cstring>
#include <fstream>
#include <stdexcept>

namespace w2x {
namespace {

struct Span {
    const uint8_t* p;
    const uint8_t* e;
    bool done() const { return p >= e; }
};

uint64_t varint(Span& s) {
    uint64_t r = 0;
    int shift = 0;
    while (true) {
        if (s.p >= s.e) throw std::runtime_error("onnx: truncated varint");
        uint8_t b = *s.p++;
        r |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return r;
        shift += 7;
        if (shift > 63) throw std::runtime_error("onnx: varint too long");
    }
}

struct Field {
    int num, wt;
    uint64_t v;  // varint value, or fixed bits
    Span sub;    // length-delimited payload
};

bool next(Span& s, Field& f) {
    if (s.done()) return false;
    uint64_t key = varint(s);
    f.num = (int)(key >> 3);
    f.wt = (int)(key & 7);
    switch (f.wt) {
        case 0: f.v = varint(s); break;
        case 1:
            if (s.e - s.p < 8) throw std::runtime_error("onnx: truncated fixed64");
            memcpy(&f.v, s.p, 8); s.p += 8; break;
        case 2: {
            uint64_t n = varint(s);
            if ((uint64_t)(s.e - s.p) < n) throw std::runtime_error("onnx: truncated bytes field");
            f.sub = Span{s.p, s.p + n}; s.p += n; break;
        }
        case 5: {
            if (s.e - s.p < 4) throw std::runtime_error("onnx: truncated fixed32");
            uint32_t t; memcpy(&t, s.p, 4); f.v = t; s.p += 4; break;
        }
        default: throw std::runtime_error("onnx: unsupported wire type");
    }
    return true;
}

std::string str(const Span& s) { return std::string((const char*)s.p, (size_t)(s.e - s.p)); }

// repeated int64: packed blob or one varint per element (torch writes unpacked)
void rep_i64(const Field& f, std::vector<int64_t>& out) {
    if (f.wt == 0) { out.push_back((int64_t)f.v); return; }
    Span s = f.sub;
    while (!s.done()) out.push_back((int64_t)varint(s));
}
void rep_f32(const Field& f, std::vector<float>& out) {
    if (f.wt == 5) { uint32_t b = (uint32_t)f.v; float x; memcpy(&x, &b, 4); out.push_back(x); return; }
    for (const uint8_t* p = f.sub.p; p + 4 <= f.sub.e; p += 4) { float x; memcpy(&x, p, 4); out.push_back(x); }
}

float half_to_float(uint16_t h) {
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 0x1F, m = h & 0x3FF, o;
    if (e == 0) {
        if (m == 0) o = s << 31;
        else { e = 127 - 15 + 1; while (!(m & 0x400)) { m <<= 1; --e; } m &= 0x3FF; o = (s << 31) | (e << 23) | (m << 13); }
    } else if (e == 31) o = (s << 31) | 0x7F800000u | (m << 13);
    else o = (s << 31) | ((e + 127 - 15) << 23) | (m << 13);
    float f; memcpy(&f, &o, 4); return f;
}

HTensorP parse_tensor(Span s, std::string* name_out) {
    auto t = std::make_shared<HTensor>();
    std::vector<float> f32; std::vector<int64_t> i32, i64; std::vector<double> f64;
    Span raw{nullptr, nullptr}; bool has_raw = false;
    Field f;
    while (next(s, f)) {
        switch (f.num) {
            case 1: rep_i64(f, t->shape); break;
            case 2: t->dtype = (int)f.v; break;
            case 4: rep_f32(f, f32); break;
            case 5: rep_i64(f, i32); break;
            case 7: rep_i64(f, i64); break;
            case 8: if (name_out) *name_out = str(f.sub); break;
            case 9: raw = f.sub; has_raw = true; break;
            case 10:
                if (f.wt == 1) { double d; memcpy(&d, &f.v, 8); f64.push_back(d); }
                else for (const uint8_t* p = f.sub.p; p + 8 <= f.sub.e; p += 8) { double d; memcpy(&d, p, 8); f64.push_back(d); }
                break;
            case 14: if (f.v == 1) throw std::runtime_error("onnx: external tensor data is not supported"); break;
            default: break;
        }
    }
    // dims come from the file: every one non-negative, the product bounded, and checked against the payload that is actually there BEFORE anything
    // of that size is allocated
    int64_t n = 1;
    for (int64_t d : t->shape) {
        if (d < 0) throw std::runtime_error("onnx: tensor with a negative dimension");
        if (d != 0 && n > (int64_t(1) << 40) / d) throw std::runtime_error("onnx: tensor dimensions overflow");
        n *= d;
    }
    auto need = [&](size_t have, size_t elt) { if ((int64_t)(have / elt) != n) throw std::runtime_error("onnx: tensor payload size mismatch"); };
    switch (t->dtype) {
        case DT_F32:
            if (has_raw) { need(raw.e - raw.p, 4); t->f.resize(n); memcpy(t->f.data(), raw.p, n * 4); }
            else { need(f32.size(), 1); t->f = std::move(f32); }
            break;
        case DT_F16:
            if (has_raw) need(raw.e - raw.p, 2); else need(i32.size(), 1);
            t->f.resize(n);
            if (has_raw) { need(raw.e - raw.p, 2); for (int64_t k = 0; k < n; ++k) { uint16_t h; memcpy(&h, raw.p + 2 * k, 2); t->f[k] = half_to_float(h); } }
            else { need(i32.size(), 1); for (int64_t k = 0; k < n; ++k) t->f[k] = half_to_float((uint16_t)i32[k]); }
            t->dtype = DT_F32;
            break;
        case DT_F64:
            if (has_raw) need(raw.e - raw.p, 8); else need(f64.size(), 1);
            t->f.resize(n);
            if (has_raw) { need(raw.e - raw.p, 8); for (int64_t k = 0; k < n; ++k) { double d; memcpy(&d, raw.p + 8 * k, 8); t->f[k] = (float)d; } }
            else { need(f64.size(), 1); for (int64_t k = 0; k < n; ++k) t->f[k] = (float)f64[k]; }
            t->dtype = DT_F32;
            break;
        case DT_I64:
            if (has_raw) { need(raw.e - raw.p, 8); t->i.resize(n); memcpy(t->i.data(), raw.p, n * 8); }
            else { need(i64.size(), 1); t->i = std::move(i64); }
            break;
        case DT_I32:
            if (has_raw) need(raw.e - raw.p, 4); else need(i32.size(), 1);
            t->i.resize(n);
            if (has_raw) { need(raw.e - raw.p, 4); for (int64_t k = 0; k < n; ++k) { int32_t v; memcpy(&v, raw.p + 4 * k, 4); t->i[k] = v; } }
            else { need(i32.size(), 1); for (int64_t k = 0; k < n; ++k) t->i[k] = (int32_t)i32[k]; }
            break;
        case DT_BOOL: case DT_U8: case DT_I8:
            if (has_raw) need(raw.e - raw.p, 1); else need(i32.size(), 1);
            t->i.resize(n);
            if (has_raw) { need(raw.e - raw.p, 1); for (int64_t k = 0; k < n; ++k) t->i[k] = t->dtype == DT_I8 ? (int64_t)(int8_t)raw.p[k] : (int64_t)raw.p[k]; }
            else { need(i32.size(), 1); for (int64_t k = 0; k < n; ++k) t->i[k] = i32[k]; }
            break;
        default: throw std::runtime_error("onnx: unsupported tensor data type " + std::to_string(t->dtype));
    }
    return t;
}

Attr parse_attr(Span s, std::string& name) {
    Attr a; Field f;
    bool has_f = false, has_i = false, has_s = false;
    while (next(s, f)) {
        switch (f.num) {
            case 1: name = str(f.sub); break;
            case 2: { uint32_t b = (uint32_t)f.v; memcpy(&a.f, &b, 4); has_f = true; break; }
            case 3: a.i = (int64_t)f.v; has_i = true; break;
            case 4: a.s = str(f.sub); has_s = true; break;
            case 5: a.t = parse_tensor(f.sub, nullptr); break;
            case 7: rep_f32(f, a.floats); break;
            case 8: rep_i64(f, a.ints); break;
            case 20: a.type = (int)f.v; break;
            default: break;
        }
    }
    if (a.type == 0) a.type = a.t ? 4 : !a.ints.empty() ? 7 : !a.floats.empty() ? 6 : has_i ? 2 : has_f ? 1 : has_s ? 3 : 0;
    return a;
}

Node parse_node(Span s) {
    Node n; Field f;
    while (next(s, f)) {
        switch (f.num) {
            case 1: n.in.push_back(str(f.sub)); break;
            case 2: n.out.push_back(str(f.sub)); break;
            case 3: n.name = str(f.sub); break;
            case 4: n.op = str(f.sub); break;
            case 5: { std::string k; Attr a = parse_attr(f.sub, k); n.attr[k] = std::move(a); break; }
            default: break;
        }
    }
    return n;
}

ValueInfo parse_value_info(Span s) {
    ValueInfo vi; Field f;
    while (next(s, f)) {
        if (f.num == 1) vi.name = str(f.sub);
        else if (f.num == 2) {  // TypeProto
            Span t = f.sub; Field g;
            while (next(t, g)) if (g.num == 1) {  // tensor_type
                Span tt = g.sub; Field h;
                while (next(tt, h)) {
                    if (h.num == 1) vi.elem_type = (int)h.v;
                    else if (h.num == 2) {  // shape
                        Span sh = h.sub; Field d;
                        while (next(sh, d)) if (d.num == 1) {
                            Span dm = d.sub; Field x; int64_t val = -1;
                            while (next(dm, x)) if (x.num == 1) val = (int64_t)x.v;
                            vi.dims.push_back(val);
                        }
                    }
                }
            }
        }
    }
    return vi;
}

}  // namespace

Model load_onnx(const std::string& path) {
    std::ifstream file(path, std::ios::binary | std::ios::ate);
    if (!file.is_open()) throw std::runtime_error("could not open model \"" + path + "\"");
    std::streamsize size = file.tellg();
    std::vector<uint8_t> buf((size_t)size);
    file.seekg(0, std::ios::beg);
    file.read((char*)buf.data(), size);
    Model m;
    Span s{buf.data(), buf.data() + buf.size()};
    Field f;
    bool have_graph = false;
    while (next(s, f)) {
        if (f.num == 1) m.ir_version = (int64_t)f.v;
        else if (f.num == 2) m.producer = str(f.sub);
        else if (f.num == 7) {
            have_graph = true;
            Span g = f.sub; Field h;
            while (next(g, h)) {
                if (h.num == 1) m.nodes.push_back(parse_node(h.sub));
                else if (h.num == 5) { std::string name; auto t = parse_tensor(h.sub, &name); m.init[name] = t; }
                else if (h.num == 11) m.inputs.push_back(parse_value_info(h.sub));
                else if (h.num == 12) m.outputs.push_back(parse_value_info(h.sub));
            }
        } else if (f.num == 8) {
            Span o = f.sub; Field h; std::string dom; int64_t ver = 0;
            while (next(o, h)) { if (h.num == 1) dom = str(h.sub); else if (h.num == 2) ver = (int64_t)h.v; }
            if (dom.empty() || dom == "ai.onnx") m.opset = ver;
        }
    }
    if (!have_graph) throw std::runtime_error("onnx: file has no graph");
    // graph inputs that are really initialisers (IR < 4 style) are not runtime inputs
    std::vector<ValueInfo> real;
    for (auto& vi : m.inputs) if (!m.init.count(vi.name)) real.push_back(vi);
    m.inputs = real;
    return m;
}

}  // namespace w2x
