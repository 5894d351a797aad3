// ORACLE (test infrastructure, not product code): a second, loop-level fp32 CPU executor for ONNX graphs, C++ / OpenMP.
//
// What the reference delegates to TensorRT - IExecutionContext::enqueueV3 (src/tensorrt/img2img_infer.cpp:80) on an engine
// built from the ONNX file (src/tensorrt/img2img_build.cpp:54-173) - restated from the ONNX operator specification (opsets
// 11-17) as plain loops over NCHW float32 tensors: every Conv / MatMul / Softmax / LayerNormalization ... is the definition
// written out, no library underneath.  oracle/onnx_exec.py states the same graphs on torch-CPU operators; the two executors
// share nothing but the ONNX file (this file has its own protobuf reader) and tests/test_oracle_cnet.py holds them against each
// other on every graph family.  "Parity unpinned" against TensorRT itself applies to both (no TensorRT here, no golden outputs
// in the reference: SURVEY 8c); what the pair pins is that the checker's reading of the graph does not hang on one
// implementation.  It is also SURVEY 8d's "C++/OpenMP fp32 CPU oracle": bench.py's cpu_baseline leg times it beside the
// torch executor.
//
// Only tests/, __graft_entry__ (build / smoke) and bench.py's cpu_baseline leg may load the library built from this file.
//
// C ABI: onet_load(path, err, cap) -> handle | null;  onet_run(handle, x, n, c, h, w, y, ycap, yshape[4], threads, err, cap) -> elements | -1;
//        onet_flops(handle) -> 2 * MACs of the Conv / ConvTranspose / MatMul / Gemm nodes of the last run;  onet_free(handle).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include <omp.h>

namespace {

using i64 = int64_t;
[[noreturn]] void fail(const std::string& m) { throw std::runtime_error(m); }

// ---------------------------------------------------------------------------------------------------------------- protobuf
struct PB {
    const uint8_t* p; const uint8_t* e;
    bool more() const { return p < e; }
    uint64_t varint() {
        uint64_t v = 0; int sh = 0;
        while (true) {
            if (p >= e || sh > 63) fail("onnx: truncated varint");
            const uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; sh += 7;
            if (!(b & 0x80)) return v;
        }
    }
    PB bytes() { const uint64_t n = varint(); if (n > (uint64_t)(e - p)) fail("onnx: truncated field"); PB s{p, p + n}; p += n; return s; }
    std::string str() { PB s = bytes(); return std::string((const char*)s.p, (size_t)(s.e - s.p)); }
    uint32_t fixed32() { if (e - p < 4) fail("onnx: truncated fixed32"); uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t fixed64() { if (e - p < 8) fail("onnx: truncated fixed64"); uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    void skip(int wire) {
        if (wire == 0) (void)varint(); else if (wire == 1) (void)fixed64(); else if (wire == 2) (void)bytes(); else if (wire == 5) (void)fixed32(); else fail("onnx: unsupported wire type");
    }
};

// ------------------------------------------------------------------------------------------------------------------ tensors
enum DT { F32, I64, B8 };      // bool values live in `i` as 0 / 1
struct Tensor {
    DT dt = F32;
    std::vector<i64> shape;
    std::vector<float> f;
    std::vector<i64> i;
    i64 numel() const { i64 n = 1; for (i64 d : shape) n *= d; return n; }
    bool is_f() const { return dt == F32; }
    void alloc() { if (dt == F32) f.assign((size_t)numel(), 0.f); else i.assign((size_t)numel(), 0); }
};
using TP = std::shared_ptr<Tensor>;
TP make(DT dt, std::vector<i64> shape) { auto t = std::make_shared<Tensor>(); t->dt = dt; t->shape = std::move(shape); for (i64 d : t->shape) if (d < 0) fail("negative dimension"); t->alloc(); return t; }
std::vector<i64> strides_of(const std::vector<i64>& s) { std::vector<i64> st(s.size(), 1); for (int d = (int)s.size() - 2; d >= 0; --d) st[d] = st[d + 1] * s[d + 1]; return st; }
std::vector<i64> ints_of(const Tensor& t) { if (t.dt == F32) fail("integer tensor expected"); return t.i; }
float half_to_float(uint16_t h) {
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023; uint32_t u;
    if (e == 0) { if (!m) u = s << 31; else { int k = 0; uint32_t mm = m; while (!(mm & 1024)) { mm <<= 1; ++k; } u = (s << 31) | ((uint32_t)(113 - k) << 23) | ((mm & 1023) << 13); } }
    else if (e == 31) u = (s << 31) | 0x7F800000u | (m << 13);
    else u = (s << 31) | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

struct Attr { std::string name; float f = 0; i64 i = 0; std::string s; TP t; std::vector<float> floats; std::vector<i64> ints; bool has_f = false, has_i = false; };
struct Node { std::string op, name; std::vector<std::string> in, out; std::map<std::string, Attr> at; };

TP parse_tensor(PB b, std::string* name_out) {
    std::vector<i64> dims; int dtp = 0; std::vector<float> fd; std::vector<i64> id; std::vector<double> dd; PB raw{nullptr, nullptr}; bool has_raw = false;
    while (b.more()) {
        const uint64_t k = b.varint(); const int fld = (int)(k >> 3), w = (int)(k & 7);
        if (fld == 1) { if (w == 2) { PB s = b.bytes(); while (s.more()) dims.push_back((i64)s.varint()); } else dims.push_back((i64)b.varint()); }
        else if (fld == 2) dtp = (int)b.varint();
        else if (fld == 4) { if (w == 2) { PB s = b.bytes(); while (s.more()) { uint32_t u = s.fixed32(); float v; memcpy(&v, &u, 4); fd.push_back(v); } } else { uint32_t u = b.fixed32(); float v; memcpy(&v, &u, 4); fd.push_back(v); } }
        else if (fld == 5 || fld == 7) { if (w == 2) { PB s = b.bytes(); while (s.more()) id.push_back((i64)s.varint()); } else id.push_back((i64)b.varint()); }
        else if (fld == 10) { if (w == 2) { PB s = b.bytes(); while (s.more()) { uint64_t u = s.fixed64(); double v; memcpy(&v, &u, 8); dd.push_back(v); } } else { uint64_t u = b.fixed64(); double v; memcpy(&v, &u, 8); dd.push_back(v); } }
        else if (fld == 8) { std::string n = b.str(); if (name_out) *name_out = n; }
        else if (fld == 9) { raw = b.bytes(); has_raw = true; }
        else b.skip(w);
    }
    i64 n = 1; for (i64 d : dims) { if (d < 0 || (d && n > ((i64)1 << 40) / d)) fail("onnx: tensor too large"); n *= d; }
    const DT dt = dtp == 1 || dtp == 10 || dtp == 11 ? F32 : dtp == 9 ? B8 : (dtp == 6 || dtp == 7 || dtp == 2 || dtp == 3) ? I64 : (fail("onnx: unsupported tensor data type " + std::to_string(dtp)), F32);
    TP t = make(dt, dims);
    const size_t rawn = has_raw ? (size_t)(raw.e - raw.p) : 0;
    auto need = [&](size_t bytes) { if (rawn < bytes) fail("onnx: raw_data shorter than the tensor"); };
    if (dtp == 1) { if (has_raw) { need((size_t)n * 4); memcpy(t->f.data(), raw.p, (size_t)n * 4); } else { if ((i64)fd.size() < n) fail("onnx: float_data shorter than the tensor"); std::copy(fd.begin(), fd.begin() + n, t->f.begin()); } }
    else if (dtp == 10) { if (has_raw) { need((size_t)n * 2); for (i64 k = 0; k < n; ++k) { uint16_t h; memcpy(&h, raw.p + 2 * k, 2); t->f[k] = half_to_float(h); } } else { if ((i64)id.size() < n) fail("onnx: int32_data shorter than the tensor"); for (i64 k = 0; k < n; ++k) t->f[k] = half_to_float((uint16_t)id[k]); } }
    else if (dtp == 11) { if (has_raw) { need((size_t)n * 8); for (i64 k = 0; k < n; ++k) { double v; memcpy(&v, raw.p + 8 * k, 8); t->f[k] = (float)v; } } else { if ((i64)dd.size() < n) fail("onnx: double_data shorter than the tensor"); for (i64 k = 0; k < n; ++k) t->f[k] = (float)dd[k]; } }
    else if (dtp == 7) { if (has_raw) { need((size_t)n * 8); memcpy(t->i.data(), raw.p, (size_t)n * 8); } else { if ((i64)id.size() < n) fail("onnx: int64_data shorter than the tensor"); std::copy(id.begin(), id.begin() + n, t->i.begin()); } }
    else if (dtp == 6) { if (has_raw) { need((size_t)n * 4); for (i64 k = 0; k < n; ++k) { int32_t v; memcpy(&v, raw.p + 4 * k, 4); t->i[k] = v; } } else { if ((i64)id.size() < n) fail("onnx: int32_data shorter than the tensor"); for (i64 k = 0; k < n; ++k) t->i[k] = (int32_t)id[k]; } }
    else { if (has_raw) { need((size_t)n); for (i64 k = 0; k < n; ++k) t->i[k] = dtp == 3 ? (i64)(int8_t)raw.p[k] : (i64)raw.p[k]; } else { if ((i64)id.size() < n) fail("onnx: int32_data shorter than the tensor"); for (i64 k = 0; k < n; ++k) t->i[k] = id[k]; } if (dtp == 9) for (auto& v : t->i) v = v != 0; }
    return t;
}

Attr parse_attr(PB b) {
    Attr a;
    while (b.more()) {
        const uint64_t k = b.varint(); const int fld = (int)(k >> 3), w = (int)(k & 7);
        if (fld == 1) a.name = b.str();
        else if (fld == 2) { uint32_t u = b.fixed32(); memcpy(&a.f, &u, 4); a.has_f = true; }
        else if (fld == 3) { a.i = (i64)b.varint(); a.has_i = true; }
        else if (fld == 4) a.s = b.str();
        else if (fld == 5) a.t = parse_tensor(b.bytes(), nullptr);
        else if (fld == 7) { if (w == 2) { PB s = b.bytes(); while (s.more()) { uint32_t u = s.fixed32(); float v; memcpy(&v, &u, 4); a.floats.push_back(v); } } else { uint32_t u = b.fixed32(); float v; memcpy(&v, &u, 4); a.floats.push_back(v); } }
        else if (fld == 8) { if (w == 2) { PB s = b.bytes(); while (s.more()) a.ints.push_back((i64)s.varint()); } else a.ints.push_back((i64)b.varint()); }
        else b.skip(w);
    }
    return a;
}

Node parse_node(PB b) {
    Node n;
    while (b.more()) {
        const uint64_t k = b.varint(); const int fld = (int)(k >> 3), w = (int)(k & 7);
        if (fld == 1) n.in.push_back(b.str());
        else if (fld == 2) n.out.push_back(b.str());
        else if (fld == 3) n.name = b.str();
        else if (fld == 4) n.op = b.str();
        else if (fld == 5) { Attr a = parse_attr(b.bytes()); n.at[a.name] = a; }
        else b.skip(w);
    }
    return n;
}

struct Net {
    std::vector<Node> nodes;
    std::map<std::string, TP> consts;      // initializers + everything that does not depend on the graph input (folded by the first run)
    std::string input, output;
    std::vector<char> is_static;           // per node, decided by the first run
    bool folded = false;
    double flops = 0;
};

std::string value_name(PB b) { std::string n; while (b.more()) { const uint64_t k = b.varint(); if ((k >> 3) == 1) n = b.str(); else b.skip((int)(k & 7)); } return n; }

void parse_graph(PB b, Net& net) {
    std::vector<std::string> inputs, outputs;
    while (b.more()) {
        const uint64_t k = b.varint(); const int fld = (int)(k >> 3), w = (int)(k & 7);
        if (fld == 1) net.nodes.push_back(parse_node(b.bytes()));
        else if (fld == 5) { std::string name; TP t = parse_tensor(b.bytes(), &name); net.consts[name] = t; }
        else if (fld == 11) inputs.push_back(value_name(b.bytes()));
        else if (fld == 12) outputs.push_back(value_name(b.bytes()));
        else b.skip(w);
    }
    for (const std::string& s : inputs) if (!net.consts.count(s)) { if (!net.input.empty()) fail("onnx: more than one graph input"); net.input = s; }
    if (net.input.empty() || outputs.size() != 1) fail("onnx: expected one graph input and one output");
    net.output = outputs[0];
}

// ------------------------------------------------------------------------------------------------------------- N-d helpers
std::vector<i64> bshape(const std::vector<i64>& a, const std::vector<i64>& b) {
    const size_t n = std::max(a.size(), b.size()); std::vector<i64> o(n);
    for (size_t d = 0; d < n; ++d) {
        const i64 x = d + a.size() >= n ? a[d + a.size() - n] : 1, y = d + b.size() >= n ? b[d + b.size() - n] : 1;
        if (x != y && x != 1 && y != 1) fail("shapes do not broadcast");
        o[d] = x == 1 ? y : x;
    }
    return o;
}
// strides of `s` seen through output shape `o` (0 where broadcast)
std::vector<i64> bstrides(const std::vector<i64>& s, const std::vector<i64>& o) {
    const std::vector<i64> st = strides_of(s); std::vector<i64> r(o.size(), 0);
    for (size_t d = 0; d < s.size(); ++d) { const size_t od = d + o.size() - s.size(); r[od] = s[d] == 1 ? 0 : st[d]; }
    return r;
}
// threads for a loop of `work` elementary steps: a team no larger than the work feeds (a fork / join over 128 threads costs more than a small operator)
int team(i64 work) { const i64 t = work / 32768; return (int)std::max<i64>(1, std::min<i64>(t, omp_get_max_threads())); }
// rows of an N-d index space: fn(row, offsets...) with the last dimension left to the caller
template <class Fn> void for_rows(const std::vector<i64>& shape, const std::vector<std::vector<i64>>& strides, Fn fn) {
    const int nd = (int)shape.size();
    const i64 L = nd ? shape[nd - 1] : 1; i64 rows = 1; for (int d = 0; d + 1 < nd; ++d) rows *= shape[d];
    if (L == 0 || rows == 0) return;
    const int ns = (int)strides.size();
#pragma omp parallel for schedule(static) num_threads(team(rows * L))
    for (i64 r = 0; r < rows; ++r) {
        i64 off[4] = {0, 0, 0, 0}; i64 rem = r;
        for (int d = nd - 2; d >= 0; --d) { const i64 ix = rem % shape[d]; rem /= shape[d]; for (int s = 0; s < ns; ++s) off[s] += ix * strides[s][d]; }
        fn(r, L, off);
    }
}

template <class T> const std::vector<T>& data_of(const Tensor& t);
template <> const std::vector<float>& data_of<float>(const Tensor& t) { return t.f; }
template <> const std::vector<i64>& data_of<i64>(const Tensor& t) { return t.i; }
template <class T> std::vector<T>& data_of(Tensor& t);
template <> std::vector<float>& data_of<float>(Tensor& t) { return t.f; }
template <> std::vector<i64>& data_of<i64>(Tensor& t) { return t.i; }

TP to_float(const TP& t) { if (t->dt == F32) return t; TP o = make(F32, t->shape); for (size_t k = 0; k < o->f.size(); ++k) o->f[k] = (float)t->i[k]; return o; }

template <class T, class R, class Op> TP binary_t(const Tensor& a, const Tensor& b, DT odt, Op op) {
    const std::vector<i64> os = bshape(a.shape, b.shape);
    TP o = make(odt, os);
    const std::vector<T>& A = data_of<T>(a); const std::vector<T>& B = data_of<T>(b); std::vector<R>& O = data_of<R>(*o);
    const std::vector<i64> sa = bstrides(a.shape, os), sb = bstrides(b.shape, os);
    const int nd = (int)os.size(); const i64 la = nd ? sa[nd - 1] : 0, lb = nd ? sb[nd - 1] : 0;
    for_rows(os, {sa, sb}, [&](i64 r, i64 L, const i64* off) {
        const T* pa = A.data() + off[0]; const T* pb = B.data() + off[1]; R* po = O.data() + r * L;
        if (la == 1 && lb == 1) for (i64 j = 0; j < L; ++j) po[j] = op(pa[j], pb[j]);
        else if (la == 1 && lb == 0) { const T y = pb[0]; for (i64 j = 0; j < L; ++j) po[j] = op(pa[j], y); }
        else if (la == 0 && lb == 1) { const T x = pa[0]; for (i64 j = 0; j < L; ++j) po[j] = op(x, pb[j]); }
        else for (i64 j = 0; j < L; ++j) po[j] = op(pa[j * la], pb[j * lb]);
    });
    return o;
}
template <class FOp, class IOp> TP arith(const TP& a, const TP& b, FOp fop, IOp iop) {
    if (a->is_f() || b->is_f()) { TP x = to_float(a), y = to_float(b); return binary_t<float, float>(*x, *y, F32, fop); }
    return binary_t<i64, i64>(*a, *b, a->dt == B8 && b->dt == B8 ? B8 : I64, iop);
}
template <class FOp, class IOp> TP compare(const TP& a, const TP& b, FOp fop, IOp iop) {
    if (a->is_f() || b->is_f()) { TP x = to_float(a), y = to_float(b); return binary_t<float, i64>(*x, *y, B8, fop); }
    return binary_t<i64, i64>(*a, *b, B8, iop);
}
template <class Op> TP unary_f(const TP& a, Op op) {
    if (!a->is_f()) fail("float tensor expected");
    TP o = make(F32, a->shape); const i64 n = a->numel(); const float* x = a->f.data(); float* y = o->f.data();
#pragma omp parallel for schedule(static) num_threads(team(n))
    for (i64 k = 0; k < n; ++k) y[k] = op(x[k]);
    return o;
}
// out[index] = in[base + sum index[d] * stride[d]]: Transpose, Slice, Expand
TP strided_copy(const Tensor& a, const std::vector<i64>& oshape, const std::vector<i64>& st, i64 base) {
    TP o = make(a.dt, oshape);
    const int nd = (int)oshape.size(); const i64 ls = nd ? st[nd - 1] : 0;
    if (a.dt == F32) { const float* A = a.f.data(); float* O = o->f.data(); for_rows(oshape, {st}, [&](i64 r, i64 L, const i64* off) { const float* p = A + base + off[0]; float* q = O + r * L; if (ls == 1) memcpy(q, p, (size_t)L * 4); else for (i64 j = 0; j < L; ++j) q[j] = p[j * ls]; }); }
    else { const i64* A = a.i.data(); i64* O = o->i.data(); for_rows(oshape, {st}, [&](i64 r, i64 L, const i64* off) { const i64* p = A + base + off[0]; i64* q = O + r * L; for (i64 j = 0; j < L; ++j) q[j] = p[j * ls]; }); }
    return o;
}
TP reshaped(const TP& a, std::vector<i64> shape) { i64 n = 1; for (i64 d : shape) n *= d; if (n != a->numel()) fail("Reshape: element count changes"); TP o = std::make_shared<Tensor>(*a); o->shape = std::move(shape); return o; }
i64 norm_axis(i64 ax, i64 nd) { if (ax < -nd || ax >= nd) fail("axis out of range"); return ax < 0 ? ax + nd : ax; }

// ------------------------------------------------------------------------------------------------------------------ kernels
// y[j] += a * x[j]: the inner loop of the convolutions and the matrix products (clones: the library is built where it does not run)
__attribute__((target_clones("avx512f", "avx2", "default"))) void axpy(float* __restrict__ y, const float* __restrict__ x, float a, i64 n) { for (i64 j = 0; j < n; ++j) y[j] += a * x[j]; }
__attribute__((target_clones("avx512f", "avx2", "default"))) void axpy_strided(float* __restrict__ y, const float* __restrict__ x, float a, i64 n, i64 sx) { for (i64 j = 0; j < n; ++j) y[j] += a * x[j * sx]; }

struct ConvArgs { i64 sy = 1, sx = 1, pt = 0, pl = 0, pb = 0, pr = 0, dy = 1, dx = 1, group = 1, opy = 0, opx = 0; };
ConvArgs conv_args(const Node& n) {
    ConvArgs c;
    auto get = [&](const char* k) -> const std::vector<i64>* { auto it = n.at.find(k); return it == n.at.end() ? nullptr : &it->second.ints; };
    if (auto v = get("strides")) { if (v->size() != 2) fail(n.op + ": 2-d only"); c.sy = (*v)[0]; c.sx = (*v)[1]; }
    if (auto v = get("pads")) { if (v->size() != 4) fail(n.op + ": 2-d only"); c.pt = (*v)[0]; c.pl = (*v)[1]; c.pb = (*v)[2]; c.pr = (*v)[3]; }
    if (auto v = get("dilations")) { if (v->size() != 2) fail(n.op + ": 2-d only"); c.dy = (*v)[0]; c.dx = (*v)[1]; }
    if (auto v = get("output_padding")) { if (v->size() != 2) fail(n.op + ": 2-d only"); c.opy = (*v)[0]; c.opx = (*v)[1]; }
    auto g = n.at.find("group"); if (g != n.at.end()) c.group = g->second.i;
    auto ap = n.at.find("auto_pad"); if (ap != n.at.end() && !ap->second.s.empty() && ap->second.s != "NOTSET") fail(n.op + ": auto_pad is not supported");
    if (c.sy <= 0 || c.sx <= 0 || c.dy <= 0 || c.dx <= 0 || c.group <= 0) fail(n.op + ": bad attributes");
    return c;
}

TP conv2d(const Tensor& x, const Tensor& w, const Tensor* bias, const ConvArgs& c, double& flops) {
    if (x.shape.size() != 4 || w.shape.size() != 4 || !x.is_f() || !w.is_f()) fail("Conv: 4-d float tensors expected");
    const i64 N = x.shape[0], C = x.shape[1], H = x.shape[2], W = x.shape[3], OC = w.shape[0], CG = w.shape[1], KH = w.shape[2], KW = w.shape[3];
    if (C != CG * c.group || OC % c.group) fail("Conv: channel counts do not match");
    const i64 OH = (H + c.pt + c.pb - c.dy * (KH - 1) - 1) / c.sy + 1, OW = (W + c.pl + c.pr - c.dx * (KW - 1) - 1) / c.sx + 1;
    if (OH <= 0 || OW <= 0) fail("Conv: empty output");
    if (bias && (bias->numel() != OC || !bias->is_f())) fail("Conv: bias shape");
    TP o = make(F32, {N, OC, OH, OW});
    const i64 ocg = OC / c.group;
    const float* X = x.f.data(); const float* Wt = w.f.data(); float* O = o->f.data();
#pragma omp parallel for collapse(2) schedule(static) num_threads(team(N * OC * OH * OW * CG * KH * KW / 8))
    for (i64 nb = 0; nb < N * OC; ++nb)
        for (i64 oy = 0; oy < OH; ++oy) {
            const i64 n = nb / OC, oc = nb % OC, g = oc / ocg;
            float* row = O + ((n * OC + oc) * OH + oy) * OW;
            const float b0 = bias ? bias->f[oc] : 0.f;
            for (i64 ox = 0; ox < OW; ++ox) row[ox] = b0;
            for (i64 ic = 0; ic < CG; ++ic)
                for (i64 ky = 0; ky < KH; ++ky) {
                    const i64 iy = oy * c.sy - c.pt + ky * c.dy;
                    if (iy < 0 || iy >= H) continue;
                    const float* xin = X + ((n * C + g * CG + ic) * H + iy) * W;
                    for (i64 kx = 0; kx < KW; ++kx) {
                        const float wv = Wt[((oc * CG + ic) * KH + ky) * KW + kx];
                        // output columns whose input column ox * sx - pl + kx * dx lies inside the row
                        const i64 shift = kx * c.dx - c.pl;
                        i64 lo = shift >= 0 ? 0 : (-shift + c.sx - 1) / c.sx, hi = std::min(OW, (W - 1 - shift) / c.sx + 1);
                        if (W - 1 - shift < 0) hi = 0;
                        if (hi <= lo) continue;
                        if (c.sx == 1) axpy(row + lo, xin + lo + shift, wv, hi - lo);
                        else axpy_strided(row + lo, xin + lo * c.sx + shift, wv, hi - lo, c.sx);
                    }
                }
        }
    flops += 2.0 * (double)o->numel() * CG * KH * KW;
    return o;
}

TP conv_transpose2d(const Tensor& x, const Tensor& w, const Tensor* bias, const ConvArgs& c, double& flops) {
    if (x.shape.size() != 4 || w.shape.size() != 4 || !x.is_f() || !w.is_f()) fail("ConvTranspose: 4-d float tensors expected");
    if (c.group != 1 || c.dy != 1 || c.dx != 1) fail("ConvTranspose: groups / dilations are not supported");
    const i64 N = x.shape[0], C = x.shape[1], H = x.shape[2], W = x.shape[3], OC = w.shape[1], KH = w.shape[2], KW = w.shape[3];
    if (w.shape[0] != C) fail("ConvTranspose: channel counts do not match");
    const i64 OH = (H - 1) * c.sy - c.pt - c.pb + KH + c.opy, OW = (W - 1) * c.sx - c.pl - c.pr + KW + c.opx;
    if (OH <= 0 || OW <= 0) fail("ConvTranspose: empty output");
    if (bias && (bias->numel() != OC || !bias->is_f())) fail("ConvTranspose: bias shape");
    TP o = make(F32, {N, OC, OH, OW});
    const float* X = x.f.data(); const float* Wt = w.f.data(); float* O = o->f.data();
    // gathered form: out[oy][ox] = sum over (ic, ky, kx) with oy = iy * sy - pt + ky: one writer per output row
#pragma omp parallel for collapse(2) schedule(static) num_threads(team(N * OC * OH * OW * C / 8))
    for (i64 nb = 0; nb < N * OC; ++nb)
        for (i64 oy = 0; oy < OH; ++oy) {
            const i64 n = nb / OC, oc = nb % OC;
            float* row = O + ((n * OC + oc) * OH + oy) * OW;
            const float b0 = bias ? bias->f[oc] : 0.f;
            for (i64 ox = 0; ox < OW; ++ox) row[ox] = b0;
            for (i64 ky = 0; ky < KH; ++ky) {
                const i64 t = oy + c.pt - ky;
                if (t < 0 || t % c.sy) continue;
                const i64 iy = t / c.sy;
                if (iy >= H) continue;
                for (i64 ic = 0; ic < C; ++ic) {
                    const float* xin = X + ((n * C + ic) * H + iy) * W;
                    for (i64 kx = 0; kx < KW; ++kx) {
                        const float wv = Wt[((ic * OC + oc) * KH + ky) * KW + kx];
                        for (i64 ix = 0; ix < W; ++ix) { const i64 ox = ix * c.sx - c.pl + kx; if (ox >= 0 && ox < OW) row[ox] += wv * xin[ix]; }
                    }
                }
            }
        }
    flops += 2.0 * (double)x.numel() * OC * KH * KW;
    return o;
}

template <class T> TP matmul_t(const Tensor& a, const Tensor& b, double& flops) {
    std::vector<i64> as = a.shape, bs = b.shape;
    if (as.empty() || bs.empty()) fail("MatMul: scalars");
    const bool a1 = as.size() == 1, b1 = bs.size() == 1;
    if (a1) as.insert(as.begin(), 1);
    if (b1) bs.push_back(1);
    const i64 M = as[as.size() - 2], K = as[as.size() - 1], N = bs[bs.size() - 1];
    if (bs[bs.size() - 2] != K) fail("MatMul: inner dimensions differ");
    const std::vector<i64> ab(as.begin(), as.end() - 2), bb(bs.begin(), bs.end() - 2), ob = bshape(ab, bb);
    std::vector<i64> os = ob; os.push_back(M); os.push_back(N);
    TP o = make(std::is_same<T, float>::value ? F32 : I64, os);
    const std::vector<i64> sa = bstrides(ab, ob), sb = bstrides(bb, ob);
    i64 batches = 1; for (i64 d : ob) batches *= d;
    const T* A = data_of<T>(a).data(); const T* B = data_of<T>(b).data(); T* O = data_of<T>(*o).data();
    const int nb = (int)ob.size();
#pragma omp parallel for schedule(static) num_threads(team(batches * M * N * K / 8))
    for (i64 r = 0; r < batches * M; ++r) {
        const i64 bi = r / M, m = r % M;
        i64 oa = 0, obb = 0, rem = bi;
        for (int d = nb - 1; d >= 0; --d) { const i64 ix = rem % ob[d]; rem /= ob[d]; oa += ix * sa[d]; obb += ix * sb[d]; }
        const T* pa = A + oa * M * K + m * K; const T* pb = B + obb * K * N; T* po = O + r * N;
        for (i64 j = 0; j < N; ++j) po[j] = 0;
        for (i64 k = 0; k < K; ++k) {
            if constexpr (std::is_same<T, float>::value) axpy(po, pb + k * N, pa[k], N);
            else { const T x = pa[k]; for (i64 j = 0; j < N; ++j) po[j] += x * pb[k * N + j]; }
        }
    }
    if (std::is_same<T, float>::value) flops += 2.0 * (double)o->numel() * K;
    if (a1) o->shape.erase(o->shape.end() - 2);
    if (b1) o->shape.pop_back();
    return o;
}

TP softmax(const Tensor& x, i64 axis) {
    if (!x.is_f()) fail("Softmax: float tensor expected");
    const i64 nd = (i64)x.shape.size(); axis = norm_axis(axis, nd);
    i64 outer = 1, inner = 1; for (i64 d = 0; d < axis; ++d) outer *= x.shape[d]; for (i64 d = axis + 1; d < nd; ++d) inner *= x.shape[d];
    const i64 L = x.shape[axis];
    TP o = make(F32, x.shape);
    const float* X = x.f.data(); float* O = o->f.data();
#pragma omp parallel for schedule(static) num_threads(team(outer * inner * L))
    for (i64 r = 0; r < outer * inner; ++r) {
        const i64 base = (r / inner) * L * inner + r % inner;
        float mx = -INFINITY; for (i64 j = 0; j < L; ++j) mx = std::max(mx, X[base + j * inner]);
        double sum = 0; for (i64 j = 0; j < L; ++j) { const float e = std::exp(X[base + j * inner] - mx); O[base + j * inner] = e; sum += e; }
        const float inv = (float)(1.0 / sum); for (i64 j = 0; j < L; ++j) O[base + j * inner] *= inv;
    }
    return o;
}

TP layer_norm(const Tensor& x, const Tensor* scale, const Tensor* bias, i64 axis, float eps) {
    if (!x.is_f()) fail("LayerNormalization: float tensor expected");
    const i64 nd = (i64)x.shape.size(); axis = norm_axis(axis, nd);
    i64 outer = 1, L = 1; for (i64 d = 0; d < axis; ++d) outer *= x.shape[d]; for (i64 d = axis; d < nd; ++d) L *= x.shape[d];
    if ((scale && scale->numel() != L) || (bias && bias->numel() != L)) fail("LayerNormalization: scale / bias shape");
    TP o = make(F32, x.shape);
    const float* X = x.f.data(); float* O = o->f.data();
#pragma omp parallel for schedule(static) num_threads(team(outer * L))
    for (i64 r = 0; r < outer; ++r) {
        const float* p = X + r * L; float* q = O + r * L;
        double s = 0; for (i64 j = 0; j < L; ++j) s += p[j];
        const double mean = s / (double)L;
        double v = 0; for (i64 j = 0; j < L; ++j) { const double d = p[j] - mean; v += d * d; }
        const double rstd = 1.0 / std::sqrt(v / (double)L + (double)eps);
        for (i64 j = 0; j < L; ++j) { float y = (float)((p[j] - mean) * rstd); if (scale) y *= scale->f[j]; if (bias) y += bias->f[j]; q[j] = y; }
    }
    return o;
}

TP reduce_mean(const Tensor& x, std::vector<i64> axes, bool keep) {
    if (!x.is_f()) fail("ReduceMean: float tensor expected");
    const i64 nd = (i64)x.shape.size();
    std::vector<char> red(nd, axes.empty() ? 1 : 0); for (i64 a : axes) red[norm_axis(a, nd)] = 1;
    std::vector<i64> kept_shape, red_shape, kept_st, red_st; const std::vector<i64> st = strides_of(x.shape);
    for (i64 d = 0; d < nd; ++d) (red[d] ? red_shape : kept_shape).push_back(x.shape[d]), (red[d] ? red_st : kept_st).push_back(st[d]);
    i64 nk = 1, nr = 1; for (i64 d : kept_shape) nk *= d; for (i64 d : red_shape) nr *= d;
    std::vector<i64> os; for (i64 d = 0; d < nd; ++d) { if (!red[d]) os.push_back(x.shape[d]); else if (keep) os.push_back(1); }
    TP o = make(F32, os);
    const float* X = x.f.data(); float* O = o->f.data();
#pragma omp parallel for schedule(static) num_threads(team(nk * nr))
    for (i64 k = 0; k < nk; ++k) {
        i64 base = 0, rem = k; for (int d = (int)kept_shape.size() - 1; d >= 0; --d) { base += (rem % kept_shape[d]) * kept_st[d]; rem /= kept_shape[d]; }
        double s = 0;
        for (i64 r = 0; r < nr; ++r) { i64 off = base, rr = r; for (int d = (int)red_shape.size() - 1; d >= 0; --d) { off += (rr % red_shape[d]) * red_st[d]; rr /= red_shape[d]; } s += X[off]; }
        O[k] = (float)(s / (double)nr);
    }
    return o;
}

// --------------------------------------------------------------------------------------------------------------- the nodes
const Attr* attr(const Node& n, const char* k) { auto it = n.at.find(k); return it == n.at.end() ? nullptr : &it->second; }
i64 attr_i(const Node& n, const char* k, i64 dflt) { const Attr* a = attr(n, k); return a && a->has_i ? a->i : dflt; }
float attr_f(const Node& n, const char* k, float dflt) { const Attr* a = attr(n, k); return a && a->has_f ? a->f : dflt; }

// Split: the one operator here with several outputs (qkv.unbind(0) exports as Split + Squeeze) - sizes from the `split` attribute (opset < 13), the second
// input (opset >= 13) or equal parts, one per output
std::vector<TP> run_split(const Node& n, const std::vector<TP>& a) {
    if (a.empty() || !a[0]) fail("Split: missing input");
    const Tensor& x = *a[0];
    const i64 nd = (i64)x.shape.size(), ax = norm_axis(attr_i(n, "axis", 0), nd);
    std::vector<i64> sizes;
    if (const Attr* s = attr(n, "split")) sizes = s->ints;
    else if (a.size() > 1 && a[1]) sizes = ints_of(*a[1]);
    else { const i64 k = (i64)n.out.size(); if (k <= 0 || x.shape[ax] % k) fail("Split: the axis does not divide evenly"); sizes.assign((size_t)k, x.shape[ax] / k); }
    i64 total = 0; for (i64 s : sizes) { if (s < 0) fail("Split: negative size"); total += s; }
    if (sizes.size() != n.out.size() || total != x.shape[ax]) fail("Split: sizes do not match the axis");
    i64 outer = 1, inner = 1; for (i64 d = 0; d < ax; ++d) outer *= x.shape[d]; for (i64 d = ax + 1; d < nd; ++d) inner *= x.shape[d];
    std::vector<TP> outs; i64 at = 0;
    for (i64 s : sizes) {
        std::vector<i64> os = x.shape; os[ax] = s;
        TP o = make(x.dt, os);
        for (i64 r = 0; r < outer; ++r) {
            if (x.is_f()) memcpy(o->f.data() + r * s * inner, x.f.data() + (r * x.shape[ax] + at) * inner, (size_t)(s * inner) * 4);
            else memcpy(o->i.data() + r * s * inner, x.i.data() + (r * x.shape[ax] + at) * inner, (size_t)(s * inner) * 8);
        }
        outs.push_back(o); at += s;
    }
    return outs;
}

TP run_node(const Node& n, const std::vector<TP>& a, double& flops) {
    const std::string& op = n.op;
    auto in = [&](size_t k) -> const TP& { if (k >= a.size() || !a[k]) fail(op + ": missing input " + std::to_string(k)); return a[k]; };
    auto opt = [&](size_t k) -> const Tensor* { return k < a.size() && a[k] ? a[k].get() : nullptr; };
    if (op == "Constant") {
        if (const Attr* v = attr(n, "value")) { if (!v->t) fail("Constant: value is not a tensor"); return v->t; }
        if (const Attr* v = attr(n, "value_float")) { TP t = make(F32, {}); t->f[0] = v->f; return t; }
        if (const Attr* v = attr(n, "value_int")) { TP t = make(I64, {}); t->i[0] = v->i; return t; }
        if (const Attr* v = attr(n, "value_ints")) { TP t = make(I64, {(i64)v->ints.size()}); t->i = v->ints; return t; }
        if (const Attr* v = attr(n, "value_floats")) { TP t = make(F32, {(i64)v->floats.size()}); t->f = v->floats; return t; }
        fail("Constant: no value");
    }
    if (op == "Identity" || op == "Dropout") return in(0);      // (Dropout at inference is the identity)
    if (op == "Conv") return conv2d(*in(0), *in(1), opt(2), conv_args(n), flops);
    if (op == "ConvTranspose") return conv_transpose2d(*in(0), *in(1), opt(2), conv_args(n), flops);
    if (op == "LeakyRelu") { const float al = attr_f(n, "alpha", 0.01f); return unary_f(in(0), [al](float x) { return x >= 0.f ? x : x * al; }); }
    if (op == "Relu") return unary_f(in(0), [](float x) { return x > 0.f ? x : 0.f; });
    if (op == "Sigmoid") return unary_f(in(0), [](float x) { return 1.f / (1.f + std::exp(-x)); });
    if (op == "Erf") return unary_f(in(0), [](float x) { return std::erf(x); });
    if (op == "Gelu") {      // opset 20: 0.5 x (1 + erf(x / sqrt 2)), or its tanh approximation
        const Attr* ap = attr(n, "approximate");
        if (ap && ap->s == "tanh") return unary_f(in(0), [](float x) { return 0.5f * x * (1.f + std::tanh(0.7978845608028654f * (x + 0.044715f * x * x * x))); });
        return unary_f(in(0), [](float x) { return 0.5f * x * (1.f + std::erf(x * 0.7071067811865476f)); });
    }
    if (op == "Sqrt") return unary_f(in(0), [](float x) { return std::sqrt(x); });
    if (op == "Exp") return unary_f(in(0), [](float x) { return std::exp(x); });
    if (op == "Tanh") return unary_f(in(0), [](float x) { return std::tanh(x); });
    if (op == "Neg") { if (in(0)->is_f()) return unary_f(in(0), [](float x) { return -x; }); TP o = make(I64, in(0)->shape); for (size_t k = 0; k < o->i.size(); ++k) o->i[k] = -in(0)->i[k]; return o; }
    if (op == "Not") { if (in(0)->dt != B8) fail("Not: boolean tensor expected"); TP o = make(B8, in(0)->shape); for (size_t k = 0; k < o->i.size(); ++k) o->i[k] = !in(0)->i[k]; return o; }
    if (op == "Add") return arith(in(0), in(1), [](float x, float y) { return x + y; }, [](i64 x, i64 y) { return x + y; });
    if (op == "Sub") return arith(in(0), in(1), [](float x, float y) { return x - y; }, [](i64 x, i64 y) { return x - y; });
    if (op == "Mul") return arith(in(0), in(1), [](float x, float y) { return x * y; }, [](i64 x, i64 y) { return x * y; });
    if (op == "Div") return arith(in(0), in(1), [](float x, float y) { return x / y; }, [](i64 x, i64 y) { return y ? x / y : (i64)0; });   // C++ integer division truncates, as ONNX asks (a zero divisor gives 0: nothing may throw inside a parallel loop)
    if (op == "Mod") {
        if (attr_i(n, "fmod", 0)) return arith(in(0), in(1), [](float x, float y) { return std::fmod(x, y); }, [](i64 x, i64 y) { return y ? x % y : (i64)0; });
        return arith(in(0), in(1), [](float x, float y) { const float r = std::fmod(x, y); return r != 0.f && ((r < 0.f) != (y < 0.f)) ? r + y : r; },
                     [](i64 x, i64 y) { if (!y) return (i64)0; const i64 r = x % y; return r != 0 && ((r < 0) != (y < 0)) ? r + y : r; });
    }
    if (op == "Pow") { TP x = to_float(in(0)), y = to_float(in(1)); return binary_t<float, float>(*x, *y, F32, [](float p, float q) { return std::pow(p, q); }); }
    if (op == "Equal") return compare(in(0), in(1), [](float x, float y) -> i64 { return x == y; }, [](i64 x, i64 y) -> i64 { return x == y; });
    if (op == "Less") return compare(in(0), in(1), [](float x, float y) -> i64 { return x < y; }, [](i64 x, i64 y) -> i64 { return x < y; });
    if (op == "Greater") return compare(in(0), in(1), [](float x, float y) -> i64 { return x > y; }, [](i64 x, i64 y) -> i64 { return x > y; });
    if (op == "Where") {
        const Tensor& c = *in(0); if (c.dt == F32) fail("Where: boolean condition expected");
        TP x = in(1), y = in(2); if (x->is_f() != y->is_f()) { x = to_float(x); y = to_float(y); }
        const std::vector<i64> os = bshape(bshape(c.shape, x->shape), y->shape);
        TP o = make(x->dt, os);
        const std::vector<i64> sc = bstrides(c.shape, os), sx = bstrides(x->shape, os), sy = bstrides(y->shape, os);
        const int nd = (int)os.size(); const i64 lc = nd ? sc[nd - 1] : 0, lx = nd ? sx[nd - 1] : 0, ly = nd ? sy[nd - 1] : 0;
        for_rows(os, {sc, sx, sy}, [&](i64 r, i64 L, const i64* off) {
            for (i64 j = 0; j < L; ++j) {
                const bool t = c.i[off[0] + j * lc] != 0;
                if (o->dt == F32) o->f[r * L + j] = t ? x->f[off[1] + j * lx] : y->f[off[2] + j * ly]; else o->i[r * L + j] = t ? x->i[off[1] + j * lx] : y->i[off[2] + j * ly];
            }
        });
        return o;
    }
    if (op == "Clip") {
        bool has_lo = false, has_hi = false; float lo = 0, hi = 0;
        if (const Tensor* t = opt(1)) { if (t->numel() != 1) fail("Clip: scalar bounds expected"); lo = t->is_f() ? t->f[0] : (float)t->i[0]; has_lo = true; } else if (const Attr* v = attr(n, "min")) { lo = v->f; has_lo = true; }
        if (const Tensor* t = opt(2)) { if (t->numel() != 1) fail("Clip: scalar bounds expected"); hi = t->is_f() ? t->f[0] : (float)t->i[0]; has_hi = true; } else if (const Attr* v = attr(n, "max")) { hi = v->f; has_hi = true; }
        return unary_f(in(0), [=](float x) { if (has_lo && !(x >= lo)) x = x != x ? x : lo; if (has_hi && !(x <= hi)) x = x != x ? x : hi; return x; });
    }
    if (op == "MatMul") { if (in(0)->is_f() != in(1)->is_f()) fail("MatMul: mixed types"); return in(0)->is_f() ? matmul_t<float>(*in(0), *in(1), flops) : matmul_t<i64>(*in(0), *in(1), flops); }
    if (op == "Gemm") {
        const Tensor& A = *in(0); const Tensor& B = *in(1);
        if (A.shape.size() != 2 || B.shape.size() != 2 || !A.is_f() || !B.is_f()) fail("Gemm: 2-d float tensors expected");
        const bool ta = attr_i(n, "transA", 0) != 0, tb = attr_i(n, "transB", 0) != 0; const float alpha = attr_f(n, "alpha", 1.f), beta = attr_f(n, "beta", 1.f);
        const i64 M = ta ? A.shape[1] : A.shape[0], K = ta ? A.shape[0] : A.shape[1], N = tb ? B.shape[0] : B.shape[1];
        if ((tb ? B.shape[1] : B.shape[0]) != K) fail("Gemm: inner dimensions differ");
        TP o = make(F32, {M, N});
#pragma omp parallel for schedule(static) num_threads(team(M * N * K / 8))
        for (i64 m = 0; m < M; ++m)
            for (i64 j = 0; j < N; ++j) {
                float s = 0.f;
                for (i64 k = 0; k < K; ++k) s += (ta ? A.f[k * M + m] : A.f[m * K + k]) * (tb ? B.f[j * K + k] : B.f[k * N + j]);
                o->f[m * N + j] = alpha * s;
            }
        flops += 2.0 * (double)M * N * K;
        if (const Tensor* C = opt(2)) { TP c = std::make_shared<Tensor>(*C); if (!c->is_f()) fail("Gemm: float C expected"); return binary_t<float, float>(*o, *c, F32, [beta](float x, float y) { return x + beta * y; }); }
        return o;
    }
    if (op == "Softmax") return softmax(*in(0), attr_i(n, "axis", -1));
    if (op == "LayerNormalization") return layer_norm(*in(0), opt(1), opt(2), attr_i(n, "axis", -1), attr_f(n, "epsilon", 1e-5f));
    if (op == "ReduceMean") {
        std::vector<i64> axes; if (const Attr* v = attr(n, "axes")) axes = v->ints; else if (const Tensor* t = opt(1)) axes = ints_of(*t);
        return reduce_mean(*in(0), axes, attr_i(n, "keepdims", 1) != 0);
    }
    if (op == "GlobalAveragePool") { if (in(0)->shape.size() != 4) fail("GlobalAveragePool: 4-d tensor expected"); return reduce_mean(*in(0), {2, 3}, true); }
    if (op == "Shape") { TP o = make(I64, {(i64)in(0)->shape.size()}); o->i = in(0)->shape; return o; }
    if (op == "Cast") {
        const i64 to = attr_i(n, "to", 0); const Tensor& x = *in(0);
        const DT dt = to == 1 || to == 10 || to == 11 ? F32 : to == 9 ? B8 : (to == 6 || to == 7 || to == 2 || to == 3) ? I64 : (fail("Cast: unsupported type " + std::to_string(to)), F32);
        TP o = make(dt, x.shape);
        for (i64 k = 0; k < x.numel(); ++k) {
            if (dt == F32) o->f[k] = x.is_f() ? x.f[k] : (float)x.i[k];
            else if (dt == B8) o->i[k] = x.is_f() ? x.f[k] != 0.f : x.i[k] != 0;
            else { i64 v = x.is_f() ? (i64)x.f[k] : x.i[k]; if (to == 6) v = (int32_t)v; else if (to == 2) v = (uint8_t)v; else if (to == 3) v = (int8_t)v; o->i[k] = v; }
        }
        return o;
    }
    if (op == "Reshape") {
        std::vector<i64> shp = ints_of(*in(1)); const Tensor& x = *in(0); i64 known = 1; int neg = -1;
        for (size_t d = 0; d < shp.size(); ++d) { if (shp[d] == 0) { if (d >= x.shape.size()) fail("Reshape: 0 beyond the input rank"); shp[d] = x.shape[d]; } if (shp[d] == -1) { if (neg >= 0) fail("Reshape: two -1"); neg = (int)d; } else known *= shp[d]; }
        if (neg >= 0) { if (!known || x.numel() % known) fail("Reshape: -1 does not divide"); shp[neg] = x.numel() / known; }
        return reshaped(in(0), shp);
    }
    if (op == "Flatten") { const Tensor& x = *in(0); const i64 ax = norm_axis(attr_i(n, "axis", 1), (i64)x.shape.size() + 1); i64 lead = 1; for (i64 d = 0; d < ax; ++d) lead *= x.shape[d]; return reshaped(in(0), {lead, lead ? x.numel() / lead : 0}); }
    if (op == "Transpose") {
        const Tensor& x = *in(0); const i64 nd = (i64)x.shape.size(); std::vector<i64> perm;
        if (const Attr* v = attr(n, "perm")) perm = v->ints; else for (i64 d = nd - 1; d >= 0; --d) perm.push_back(d);
        if ((i64)perm.size() != nd) fail("Transpose: perm rank");
        const std::vector<i64> st = strides_of(x.shape); std::vector<i64> os(nd), ost(nd);
        for (i64 d = 0; d < nd; ++d) { const i64 p = norm_axis(perm[d], nd); os[d] = x.shape[p]; ost[d] = st[p]; }
        return strided_copy(x, os, ost, 0);
    }
    if (op == "Unsqueeze") {
        std::vector<i64> axes; if (const Attr* v = attr(n, "axes")) axes = v->ints; else axes = ints_of(*in(1));
        const i64 nd = (i64)in(0)->shape.size() + (i64)axes.size(); std::vector<char> ins(nd, 0); for (i64 ax : axes) ins[norm_axis(ax, nd)] = 1;
        std::vector<i64> os; size_t src = 0; for (i64 d = 0; d < nd; ++d) os.push_back(ins[d] ? 1 : in(0)->shape[src++]);
        return reshaped(in(0), os);
    }
    if (op == "Squeeze") {
        const Tensor& x = *in(0); const i64 nd = (i64)x.shape.size(); std::vector<i64> axes; bool given = false;
        if (const Attr* v = attr(n, "axes")) { axes = v->ints; given = true; } else if (const Tensor* t = opt(1)) { axes = ints_of(*t); given = true; }
        std::vector<char> drop(nd, 0);
        if (given) for (i64 ax : axes) { const i64 d = norm_axis(ax, nd); if (x.shape[d] != 1) fail("Squeeze: dimension is not 1"); drop[d] = 1; } else for (i64 d = 0; d < nd; ++d) drop[d] = x.shape[d] == 1;
        std::vector<i64> os; for (i64 d = 0; d < nd; ++d) if (!drop[d]) os.push_back(x.shape[d]);
        return reshaped(in(0), os);
    }
    if (op == "Concat") {
        const i64 nd = (i64)in(0)->shape.size(), ax = norm_axis(attr_i(n, "axis", 0), nd);
        std::vector<i64> os = in(0)->shape; os[ax] = 0; bool any_f = false;
        for (size_t k = 0; k < a.size(); ++k) { const Tensor& t = *in(k); if ((i64)t.shape.size() != nd) fail("Concat: ranks differ"); for (i64 d = 0; d < nd; ++d) if (d != ax && t.shape[d] != os[d]) fail("Concat: shapes differ"); os[ax] += t.shape[ax]; any_f |= t.is_f(); }
        TP o = make(any_f ? F32 : in(0)->dt, os);
        i64 outer = 1, inner = 1; for (i64 d = 0; d < ax; ++d) outer *= os[d]; for (i64 d = ax + 1; d < nd; ++d) inner *= os[d];
        i64 at = 0;
        for (size_t k = 0; k < a.size(); ++k) {
            TP t = any_f ? to_float(in(k)) : in(k); const i64 len = t->shape[ax] * inner;
            for (i64 r = 0; r < outer; ++r) { if (any_f) memcpy(o->f.data() + (r * os[ax]) * inner + at, t->f.data() + r * len, (size_t)len * 4); else memcpy(o->i.data() + (r * os[ax]) * inner + at, t->i.data() + r * len, (size_t)len * 8); }
            at += len;
        }
        return o;
    }
    if (op == "Slice") {
        const Tensor& x = *in(0); const i64 nd = (i64)x.shape.size(); std::vector<i64> starts, ends, axes, steps;
        if (const Attr* v = attr(n, "starts")) { starts = v->ints; ends = attr(n, "ends") ? attr(n, "ends")->ints : std::vector<i64>(); if (const Attr* ax = attr(n, "axes")) axes = ax->ints; }   // opset < 10
        else { starts = ints_of(*in(1)); ends = ints_of(*in(2)); if (const Tensor* t = opt(3)) axes = ints_of(*t); if (const Tensor* t = opt(4)) steps = ints_of(*t); }
        if (axes.empty()) for (size_t k = 0; k < starts.size(); ++k) axes.push_back((i64)k);
        if (steps.empty()) steps.assign(starts.size(), 1);
        if (ends.size() != starts.size() || axes.size() != starts.size() || steps.size() != starts.size()) fail("Slice: argument lengths differ");
        std::vector<i64> os = x.shape, st = strides_of(x.shape); i64 base = 0;
        for (size_t k = 0; k < starts.size(); ++k) {
            const i64 ax = norm_axis(axes[k], nd), d = x.shape[ax], step = steps[k]; i64 s = starts[k], e = ends[k];
            if (!step) fail("Slice: step 0");
            if (s < 0) s += d;
            if (e < 0 && e > -((i64)1 << 62)) e += d;
            i64 len;
            if (step > 0) { s = std::max<i64>(0, std::min(d, s)); e = std::max<i64>(0, std::min(d, e)); len = e > s ? (e - s + step - 1) / step : 0; }
            else { s = std::max<i64>(-1, std::min(d - 1, s)); e = e <= -((i64)1 << 62) ? -1 : std::max<i64>(-1, std::min(d - 1, e)); len = s > e ? (s - e - step - 1) / (-step) : 0; }
            base += (len ? s : 0) * st[ax]; os[ax] = len; st[ax] *= step;
        }
        return strided_copy(x, os, st, base);
    }
    if (op == "Gather") {
        const Tensor& x = *in(0); const Tensor& idx = *in(1); if (idx.is_f()) fail("Gather: integer indices expected");
        const i64 nd = (i64)x.shape.size(), ax = norm_axis(attr_i(n, "axis", 0), nd);
        std::vector<i64> os(x.shape.begin(), x.shape.begin() + ax); os.insert(os.end(), idx.shape.begin(), idx.shape.end()); os.insert(os.end(), x.shape.begin() + ax + 1, x.shape.end());
        TP o = make(x.dt, os);
        i64 outer = 1, inner = 1; for (i64 d = 0; d < ax; ++d) outer *= x.shape[d]; for (i64 d = ax + 1; d < nd; ++d) inner *= x.shape[d];
        const i64 ni = idx.numel(), D = x.shape[ax];
        for (i64 r = 0; r < outer; ++r)
            for (i64 k = 0; k < ni; ++k) {
                i64 j = idx.i[k]; if (j < 0) j += D; if (j < 0 || j >= D) fail("Gather: index out of range");
                if (x.is_f()) memcpy(o->f.data() + (r * ni + k) * inner, x.f.data() + (r * D + j) * inner, (size_t)inner * 4); else memcpy(o->i.data() + (r * ni + k) * inner, x.i.data() + (r * D + j) * inner, (size_t)inner * 8);
            }
        return o;
    }
    if (op == "ConstantOfShape") {
        const std::vector<i64> shp = ints_of(*in(0)); const Attr* v = attr(n, "value");
        TP o = make(v && v->t ? v->t->dt : F32, shp);
        if (v && v->t) { if (v->t->numel() != 1) fail("ConstantOfShape: one value expected"); if (o->dt == F32) std::fill(o->f.begin(), o->f.end(), v->t->f[0]); else std::fill(o->i.begin(), o->i.end(), v->t->i[0]); }
        return o;
    }
    if (op == "Expand") { const Tensor& x = *in(0); const std::vector<i64> os = bshape(x.shape, ints_of(*in(1))); return strided_copy(x, os, bstrides(x.shape, os), 0); }
    if (op == "Range") {
        const Tensor& s = *in(0); const Tensor& l = *in(1); const Tensor& d = *in(2);
        if (s.is_f()) { const float a0 = s.f[0], a1 = l.f[0], st = d.f[0]; const i64 cnt = std::max<i64>(0, (i64)std::ceil((a1 - a0) / st)); TP o = make(F32, {cnt}); for (i64 k = 0; k < cnt; ++k) o->f[k] = a0 + (float)k * st; return o; }
        const i64 a0 = s.i[0], a1 = l.i[0], st = d.i[0]; if (!st) fail("Range: step 0");
        const i64 cnt = std::max<i64>(0, st > 0 ? (a1 - a0 + st - 1) / st : (a0 - a1 - st - 1) / (-st)); TP o = make(I64, {cnt}); for (i64 k = 0; k < cnt; ++k) o->i[k] = a0 + k * st; return o;
    }
    if (op == "ScatterND") {
        const Tensor& x = *in(0); const Tensor& idx = *in(1); const Tensor& upd = *in(2);
        if (idx.is_f() || idx.shape.empty() || x.is_f() != upd.is_f()) fail("ScatterND: argument types");
        const i64 k = idx.shape.back(), nd = (i64)x.shape.size(); if (k > nd) fail("ScatterND: index depth");
        const std::vector<i64> st = strides_of(x.shape); i64 inner = 1; for (i64 d = k; d < nd; ++d) inner *= x.shape[d];
        const i64 cnt = k ? idx.numel() / k : 0; if (upd.numel() != cnt * inner) fail("ScatterND: updates shape");
        TP o = std::make_shared<Tensor>(x);
        for (i64 r = 0; r < cnt; ++r) {
            i64 off = 0; for (i64 j = 0; j < k; ++j) { i64 v = idx.i[r * k + j]; if (v < 0) v += x.shape[j]; if (v < 0 || v >= x.shape[j]) fail("ScatterND: index out of range"); off += v * st[j]; }
            if (x.is_f()) memcpy(o->f.data() + off, upd.f.data() + r * inner, (size_t)inner * 4); else memcpy(o->i.data() + off, upd.i.data() + r * inner, (size_t)inner * 8);
        }
        return o;
    }
    if (op == "Pad") {
        const Tensor& x = *in(0); const i64 nd = (i64)x.shape.size(); std::vector<i64> pads;
        if (const Attr* v = attr(n, "pads")) pads = v->ints; else pads = ints_of(*in(1));
        if ((i64)pads.size() != 2 * nd) fail("Pad: pads length");
        std::string mode = "constant"; if (const Attr* v = attr(n, "mode")) mode = v->s;
        float fv = attr_f(n, "value", 0.f); i64 iv = 0;
        if (const Tensor* t = opt(2)) if (t->numel()) { if (t->is_f()) fv = t->f[0]; else iv = t->i[0]; }
        std::vector<i64> os(nd); std::vector<std::vector<i64>> map(nd);
        for (i64 d = 0; d < nd; ++d) {
            const i64 D = x.shape[d]; os[d] = D + pads[d] + pads[d + nd]; if (os[d] < 0) fail("Pad: negative size");
            map[d].resize(os[d]);
            for (i64 o = 0; o < os[d]; ++o) {
                i64 s = o - pads[d];
                if (s < 0 || s >= D) {
                    if (mode == "constant") s = -1;
                    else if (mode == "edge") s = s < 0 ? 0 : D - 1;
                    else if (mode == "reflect") { if (D < 2) fail("Pad: reflect on a dimension of 1"); const i64 p = 2 * (D - 1); s = ((s % p) + p) % p; if (s >= D) s = p - s; }
                    else fail("Pad: mode " + mode);
                }
                map[d][o] = s;
            }
        }
        TP o = make(x.dt, os);
        const std::vector<i64> st = strides_of(x.shape); const i64 total = o->numel(); const std::vector<i64> ost = strides_of(os);
#pragma omp parallel for schedule(static) num_threads(team(total))
        for (i64 k = 0; k < total; ++k) {
            i64 off = 0; bool cst = false;
            for (i64 d = 0; d < nd; ++d) { const i64 s = map[d][(k / ost[d]) % os[d]]; if (s < 0) { cst = true; break; } off += s * st[d]; }
            if (x.is_f()) o->f[k] = cst ? fv : x.f[off]; else o->i[k] = cst ? iv : x.i[off];
        }
        return o;
    }
    if (op == "DepthToSpace") {
        const Tensor& x = *in(0); if (x.shape.size() != 4 || !x.is_f()) fail("DepthToSpace: 4-d float tensor expected");
        const i64 r = attr_i(n, "blocksize", 0), B = x.shape[0], C = x.shape[1], H = x.shape[2], W = x.shape[3];
        if (r <= 0 || C % (r * r)) fail("DepthToSpace: blocksize");
        const bool crd = attr(n, "mode") && attr(n, "mode")->s == "CRD"; const i64 OC = C / (r * r);
        TP o = make(F32, {B, OC, H * r, W * r});
#pragma omp parallel for collapse(2) schedule(static) num_threads(team(x.numel()))
        for (i64 b = 0; b < B; ++b)
            for (i64 c = 0; c < OC; ++c)
                for (i64 y = 0; y < H * r; ++y)
                    for (i64 xx = 0; xx < W * r; ++xx) {
                        const i64 by = y % r, bx = xx % r, ic = crd ? c * r * r + by * r + bx : (by * r + bx) * OC + c;
                        o->f[((b * OC + c) * H * r + y) * W * r + xx] = x.f[((b * C + ic) * H + y / r) * W + xx / r];
                    }
        return o;
    }
    fail("ONNX operator " + op + " is not implemented in the C++ oracle" + (n.name.empty() ? "" : " (node " + n.name + ")"));
}

i64 run_net(Net& net, const float* x, const i64 xs[4], float* y, i64 ycap, i64 yshape[4]) {
    std::map<std::string, TP> env;
    TP xin = make(F32, {xs[0], xs[1], xs[2], xs[3]}); memcpy(xin->f.data(), x, (size_t)xin->numel() * 4);
    env[net.input] = xin;
    const size_t nn = net.nodes.size();
    if (!net.folded) {      // which nodes depend on the graph input (everything else is evaluated once and kept)
        std::set<std::string> runtime{net.input};
        net.is_static.assign(nn, 1);
        for (size_t k = 0; k < nn; ++k) {
            for (const std::string& s : net.nodes[k].in) if (!s.empty() && runtime.count(s)) net.is_static[k] = 0;
            if (!net.is_static[k]) for (const std::string& s : net.nodes[k].out) runtime.insert(s);
        }
    }
    std::map<std::string, size_t> last;
    for (size_t k = 0; k < nn; ++k) for (const std::string& s : net.nodes[k].in) if (!s.empty()) last[s] = k;
    net.flops = 0;
    for (size_t k = 0; k < nn; ++k) {
        const Node& n = net.nodes[k];
        if (net.folded && net.is_static[k]) continue;
        std::vector<TP> a;
        for (const std::string& s : n.in) {
            if (s.empty()) { a.push_back(nullptr); continue; }
            auto it = env.find(s); if (it != env.end()) { a.push_back(it->second); continue; }
            auto ic = net.consts.find(s); if (ic == net.consts.end()) fail("node " + n.name + " (" + n.op + ") reads " + s + ", which nothing produces");
            a.push_back(ic->second);
        }
        TP out;
        std::vector<TP> more;      // outputs 1.. of a Split
        try {
            if (n.op == "Split") { more = run_split(n, a); out = more[0]; more.erase(more.begin()); }
            else out = run_node(n, a, net.flops);
        } catch (const std::exception& e) { fail(std::string(e.what()) + " [node " + std::to_string(k) + " " + n.op + " " + n.name + "]"); }
        if (n.out.empty()) continue;
        if (!net.folded && net.is_static[k]) net.consts[n.out[0]] = out; else env[n.out[0]] = out;
        for (size_t j = 0; j < more.size(); ++j) { if (!net.folded && net.is_static[k]) net.consts[n.out[j + 1]] = more[j]; else env[n.out[j + 1]] = more[j]; }
        for (const std::string& s : n.in) if (!s.empty() && last[s] == k && s != net.output) env.erase(s);
    }
    net.folded = true;
    auto it = env.find(net.output); TP out = it != env.end() ? it->second : (net.consts.count(net.output) ? net.consts[net.output] : nullptr);
    if (!out || !out->is_f() || out->shape.size() != 4) fail("the graph output is not a 4-d float tensor");
    if (out->numel() > ycap) fail("output buffer too small");
    memcpy(y, out->f.data(), (size_t)out->numel() * 4);
    for (int d = 0; d < 4; ++d) yshape[d] = out->shape[d];
    return out->numel();
}

void set_err(char* err, int cap, const std::string& m) { if (err && cap > 0) { snprintf(err, (size_t)cap, "%s", m.c_str()); } }

}  // namespace

extern "C" {

void* onet_load(const char* path, char* err, int cap) {
    try {
        std::ifstream f(path, std::ios::binary); if (!f) fail(std::string("cannot open ") + path);
        std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        auto net = std::make_unique<Net>();
        PB b{buf.data(), buf.data() + buf.size()}; bool seen = false;
        while (b.more()) { const uint64_t k = b.varint(); if ((k >> 3) == 7 && (k & 7) == 2) { parse_graph(b.bytes(), *net); seen = true; } else b.skip((int)(k & 7)); }
        if (!seen) fail("onnx: no graph in the file");
        return net.release();
    } catch (const std::exception& e) { set_err(err, cap, e.what()); return nullptr; }
}

long long onet_run(void* h, const float* x, long long n, long long c, long long hh, long long w, float* y, long long ycap, long long* yshape, int threads, char* err, int cap) {
    try {
        if (!h || !x || !y || !yshape || n <= 0 || c <= 0 || hh <= 0 || w <= 0) fail("bad arguments");
        if (threads > 0) omp_set_num_threads(threads);
        const i64 xs[4] = {n, c, hh, w}; i64 ys[4];
        const i64 r = run_net(*(Net*)h, x, xs, y, ycap, ys);
        for (int d = 0; d < 4; ++d) yshape[d] = ys[d];
        return r;
    } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

double onet_flops(void* h) { return h ? ((Net*)h)->flops : 0.0; }
int onet_threads(void) { return omp_get_max_threads(); }
void onet_free(void* h) { delete (Net*)h; }

}  // extern "C"
