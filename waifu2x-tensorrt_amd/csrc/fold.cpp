#include "fold.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <stdexcept>

namespace w2x {

const Value& FoldedGraph::val(const std::string& n) const {
    auto it = vals.find(n);
    if (it == vals.end()) throw std::runtime_error("graph: unknown tensor \"" + n + "\"");
    return it->second;
}
const HTensor& FoldedGraph::cst(const std::string& n) const {
    const Value& v = val(n);
    if (!v.is_const) throw std::runtime_error("graph: tensor \"" + n + "\" is not constant");
    return *v.c;
}

namespace {

using Shape = std::vector<int64_t>;

int64_t prod(const Shape& s) { int64_t n = 1; for (auto d : s) n *= d; return n; }

Shape strides_of(const Shape& s) {
    Shape st(s.size(), 1);
    for (int k = (int)s.size() - 2; k >= 0; --k) st[k] = st[k + 1] * s[k + 1];
    return st;
}

Shape broadcast_shape(const Shape& a, const Shape& b) {
    size_t r = std::max(a.size(), b.size());
    Shape o(r);
    for (size_t k = 0; k < r; ++k) {
        int64_t da = k + a.size() >= r ? a[k + a.size() - r] : 1;
        int64_t db = k + b.size() >= r ? b[k + b.size() - r] : 1;
        if (da != db && da != 1 && db != 1) throw std::runtime_error("fold: shapes do not broadcast");
        o[k] = std::max(da, db);
        if (da == 0 || db == 0) o[k] = 0;
    }
    return o;
}

// index of the broadcast source element for flat output index
struct Bcast {
    Shape oshape, ostr, sstr;
    Bcast(const Shape& out, const Shape& src) : oshape(out), ostr(strides_of(out)), sstr(out.size(), 0) {
        Shape st = strides_of(src);
        size_t r = out.size();
        for (size_t k = 0; k < src.size(); ++k) {
            size_t ok = k + r - src.size();
            sstr[ok] = src[k] == 1 ? 0 : st[k];
        }
    }
    int64_t operator()(int64_t flat) const {
        int64_t s = 0;
        for (size_t k = 0; k < oshape.size(); ++k) { int64_t c = (flat / ostr[k]) % oshape[k]; s += c * sstr[k]; }
        return s;
    }
};

// Constants the folding pass materialises come from shapes a FILE dictates (ConstantOfShape, Expand, Range, Tile-like broadcasts ...): bounded, so that a
// model file cannot make build() allocate terabytes.  2^28 elements = 1 GiB of floats; the largest folded constant of the graphs here (a 640 x 640 tile's
// window tables and shift masks) is below 2^23.
constexpr int64_t kMaxFoldedElems = int64_t(1) << 28;
int64_t checked_numel(const Shape& shape) {
    int64_t n = 1;
    for (int64_t d : shape) {
        if (d < 0) throw std::runtime_error("fold: negative dimension " + std::to_string(d) + " in a constant's shape");
        if (d != 0 && n > kMaxFoldedElems / d) throw std::runtime_error("fold: a constant of more than 2^28 elements (shape dictated by the model file) is refused");
        n *= d;
    }
    return n;
}

HTensorP make(int dtype, const Shape& shape) {
    auto t = std::make_shared<HTensor>();
    t->dtype = dtype; t->shape = shape;
    int64_t n = checked_numel(shape);
    if (t->is_float()) t->f.assign(n, 0.f); else t->i.assign(n, 0);
    return t;
}

double getd(const HTensor& t, int64_t k) { return t.is_float() ? (double)t.f[k] : (double)t.i[k]; }

// a Transpose's perm attribute comes from the file: it must be a permutation of 0 .. rank - 1 before it indexes a shape
void check_perm(const std::vector<int64_t>& perm, int rank) {
    std::vector<char> seen((size_t)std::max(rank, 0), 0);
    bool ok = (int)perm.size() == rank;
    for (size_t k = 0; ok && k < perm.size(); ++k) { ok = perm[k] >= 0 && perm[k] < rank && !seen[(size_t)perm[k]]; if (ok) seen[(size_t)perm[k]] = 1; }
    if (!ok) throw std::runtime_error("graph: Transpose with a perm that is not a permutation of its input's axes");
}

int64_t norm_axis(int64_t ax, int rank) { if (ax < 0) ax += rank; if (ax < 0 || ax >= std::max(rank, 1)) throw std::runtime_error("fold: axis out of range"); return ax; }

HTensorP binary(const std::string& op, const HTensor& a, const HTensor& b) {
    Shape os = broadcast_shape(a.shape, b.shape);
    bool cmp = op == "Equal" || op == "Less" || op == "Greater" || op == "LessOrEqual" || op == "GreaterOrEqual";
    bool logic = op == "And" || op == "Or" || op == "Xor";
    bool fl = a.is_float() || b.is_float();
    int odt = (cmp || logic) ? DT_BOOL : (fl ? DT_F32 : a.dtype);
    auto o = make(odt, os);
    Bcast ia(os, a.shape), ib(os, b.shape);
    int64_t n = prod(os);
    for (int64_t k = 0; k < n; ++k) {
        int64_t xa = ia(k), xb = ib(k);
        if (fl) {
            float x = a.is_float() ? a.f[xa] : (float)a.i[xa], y = b.is_float() ? b.f[xb] : (float)b.i[xb];
            if (cmp) { o->i[k] = op == "Equal" ? x == y : op == "Less" ? x < y : op == "Greater" ? x > y : op == "LessOrEqual" ? x <= y : x >= y; continue; }
            float r;
            if (op == "Add") r = x + y; else if (op == "Sub") r = x - y; else if (op == "Mul") r = x * y;
            else if (op == "Div") r = x / y; else if (op == "Pow") r = std::pow(x, y);
            else if (op == "Max") r = std::max(x, y); else if (op == "Min") r = std::min(x, y);
            else throw std::runtime_error("fold: binary op " + op + " on floats");
            o->f[k] = r;
        } else {
            int64_t x = a.i[xa], y = b.i[xb];
            if (cmp) { o->i[k] = op == "Equal" ? x == y : op == "Less" ? x < y : op == "Greater" ? x > y : op == "LessOrEqual" ? x <= y : x >= y; continue; }
            int64_t r;
            if (op == "Add") r = x + y; else if (op == "Sub") r = x - y; else if (op == "Mul") r = x * y;
            else if (op == "Div") { if (y == 0) throw std::runtime_error("fold: integer division by zero"); r = x / y; }
            else if (op == "Pow") r = (int64_t)std::llround(std::pow((double)x, (double)y));
            else if (op == "Max") r = std::max(x, y); else if (op == "Min") r = std::min(x, y);
            else if (op == "And") r = x && y; else if (op == "Or") r = x || y; else if (op == "Xor") r = (x != 0) != (y != 0);
            else if (op == "Mod") r = y ? x % y : 0;
            else throw std::runtime_error("fold: binary op " + op + " on integers");
            o->i[k] = r;
        }
    }
    return o;
}

HTensorP cast_to(const HTensor& a, int to) {
    auto o = make(to == DT_F16 || to == DT_F64 ? DT_F32 : to, a.shape);
    int64_t n = a.numel();
    for (int64_t k = 0; k < n; ++k) {
        if (o->is_float()) o->f[k] = a.is_float() ? a.f[k] : (float)a.i[k];
        else if (to == DT_BOOL) o->i[k] = a.is_float() ? a.f[k] != 0.f : a.i[k] != 0;
        else o->i[k] = a.is_float() ? (int64_t)a.f[k] : a.i[k];
    }
    return o;
}

HTensorP gather_elems(const HTensor& a, const Shape& oshape, const std::function<int64_t(int64_t)>& src) {
    auto o = make(a.dtype, oshape);
    int64_t n = prod(oshape);
    for (int64_t k = 0; k < n; ++k) { int64_t s = src(k); if (a.is_float()) o->f[k] = a.f[s]; else o->i[k] = a.i[s]; }
    return o;
}

struct SliceSpec { Shape oshape, start, step; };

SliceSpec slice_spec(const Shape& in, const std::vector<int64_t>& starts, const std::vector<int64_t>& ends,
                     const std::vector<int64_t>& axes_in, const std::vector<int64_t>& steps_in) {
    int r = (int)in.size();
    SliceSpec sp; sp.oshape = in; sp.start.assign(r, 0); sp.step.assign(r, 1);
    for (size_t k = 0; k < starts.size(); ++k) {
        int64_t ax = axes_in.empty() ? (int64_t)k : norm_axis(axes_in[k], r);
        int64_t st = steps_in.empty() ? 1 : steps_in[k];
        int64_t d = in[ax], s = starts[k], e = ends[k];
        if (st == 0) throw std::runtime_error("fold: slice step 0");
        if (st > 0) {
            if (s < 0) s += d; if (e < 0) e += d;
            s = std::clamp<int64_t>(s, 0, d); e = std::clamp<int64_t>(e, 0, d);
            sp.oshape[ax] = std::max<int64_t>(0, (e - s + st - 1) / st);
        } else {
            if (s < 0) s += d;
            if (e < -d) e = -1; else if (e < 0) e += d;
            s = std::clamp<int64_t>(s, -1, d - 1); e = std::clamp<int64_t>(e, -1, d - 1);
            sp.oshape[ax] = std::max<int64_t>(0, (s - e + (-st) - 1) / (-st));
        }
        sp.start[ax] = s; sp.step[ax] = st;
    }
    return sp;
}

std::vector<int64_t> ints_of(const HTensor& t) {
    std::vector<int64_t> v(t.numel());
    for (size_t k = 0; k < v.size(); ++k) v[k] = t.is_float() ? (int64_t)t.f[k] : t.i[k];
    return v;
}

Shape reshape_target(const Shape& in, const std::vector<int64_t>& req, bool allowzero) {
    Shape o(req.size());
    int64_t known = 1; int neg = -1;
    // (the requested dims come from the file: products are formed with an overflow check, a target larger than any tensor here is a size mismatch)
    auto mul = [&](int64_t d) {
        if (d < 0) throw std::runtime_error("fold: reshape to a negative dimension");
        if (d != 0 && known > (int64_t(1) << 48) / d) throw std::runtime_error("fold: reshape size mismatch");
        known *= d;
    };
    for (size_t k = 0; k < req.size(); ++k) {
        if (req[k] == -1) { if (neg >= 0) throw std::runtime_error("fold: reshape with two -1"); neg = (int)k; o[k] = 1; }
        else if (req[k] == 0 && !allowzero) { if (k >= in.size()) throw std::runtime_error("fold: reshape 0-dim out of range"); o[k] = in[k]; mul(o[k]); }
        else { o[k] = req[k]; mul(o[k]); }
    }
    int64_t total = prod(in);
    if (neg >= 0) { if (known == 0 || total % known) throw std::runtime_error("fold: reshape size mismatch"); o[neg] = total / known; }
    else if (known != total) throw std::runtime_error("fold: reshape size mismatch");
    return o;
}

Shape unsqueeze_shape(const Shape& in, std::vector<int64_t> axes) {
    int r = (int)(in.size() + axes.size());
    for (auto& a : axes) a = norm_axis(a, r);
    std::sort(axes.begin(), axes.end());
    Shape o; size_t src = 0;
    for (int k = 0; k < r; ++k) { if (std::binary_search(axes.begin(), axes.end(), k)) o.push_back(1); else o.push_back(in[src++]); }
    return o;
}

Shape squeeze_shape(const Shape& in, std::vector<int64_t> axes, bool all) {
    int r = (int)in.size();
    for (auto& a : axes) a = norm_axis(a, r);
    Shape o;
    for (int k = 0; k < r; ++k) {
        bool drop = all ? in[k] == 1 : std::find(axes.begin(), axes.end(), k) != axes.end();
        if (drop && in[k] != 1) throw std::runtime_error("fold: squeeze of non-1 dim");
        if (!drop) o.push_back(in[k]);
    }
    return o;
}

std::vector<int64_t> axes_arg(const Node& n, const std::vector<const Value*>& in, size_t idx) {
    if (n.has("axes")) return n.aints("axes");
    if (in.size() > idx && in[idx]) {
        if (!in[idx]->is_const) throw std::runtime_error("fold: " + n.op + " axes must be constant");
        return ints_of(*in[idx]->c);
    }
    return {};
}

}  // namespace

FoldedGraph fold_graph(const Model& m, const std::vector<int64_t>& input_shape) {
    FoldedGraph g;
    g.model = &m;
    if (m.inputs.size() != 1 || m.outputs.size() != 1)
        throw std::runtime_error("model has invalid number of IO tensors: expected 2, got " + std::to_string(m.inputs.size() + m.outputs.size()));
    g.input = m.inputs[0].name; g.output = m.outputs[0].name;
    if (m.inputs[0].dims.size() != 4)
        throw std::runtime_error("model has invalid IO tensor shape: expected 4 dims, got " + std::to_string(m.inputs[0].dims.size()));
    for (size_t k = 0; k < 4; ++k) {
        int64_t d = m.inputs[0].dims[k];
        if (d > 0 && d != input_shape[k])
            throw std::runtime_error("model input dim " + std::to_string(k) + " is fixed to " + std::to_string(d) + ", configuration asks for " + std::to_string(input_shape[k]));
    }
    for (auto& kv : m.init) { Value v; v.is_const = true; v.c = kv.second; v.shape = kv.second->shape; v.dtype = kv.second->dtype; g.vals[kv.first] = v; }
    { Value v; v.shape = input_shape; v.dtype = DT_F32; g.vals[g.input] = v; }

    for (const Node& n : m.nodes) {
        std::vector<const Value*> in;
        bool all_const = true;
        for (auto& name : n.in) {
            if (name.empty()) { in.push_back(nullptr); continue; }
            auto it = g.vals.find(name);
            if (it == g.vals.end()) throw std::runtime_error("graph: node " + n.op + " reads undefined tensor \"" + name + "\"");
            in.push_back(&it->second);
            if (!it->second.is_const) all_const = false;
        }
        auto C = [&](size_t k) -> const HTensor& {
            if (k >= in.size() || !in[k] || !in[k]->is_const) throw std::runtime_error("fold: " + n.op + " needs a constant input #" + std::to_string(k));
            return *in[k]->c;
        };
        auto has_in = [&](size_t k) { return k < in.size() && in[k] != nullptr; };
        auto set_const = [&](HTensorP t) { Value v; v.is_const = true; v.c = t; v.shape = t->shape; v.dtype = t->dtype; g.vals[n.out[0]] = v; };
        auto set_dyn = [&](const Shape& s, int dt) { Value v; v.shape = s; v.dtype = dt; g.vals[n.out[0]] = v; };
        const std::string& op = n.op;

        // ---- ops that are constant even with runtime inputs
        if (op == "Shape") {
            auto t = make(DT_I64, {(int64_t)in[0]->shape.size()}); t->i = in[0]->shape; set_const(t); continue;
        }
        if (op == "Constant") {
            auto it = n.attr.find("value");
            if (it != n.attr.end() && it->second.t) { set_const(it->second.t); continue; }
            if (n.has("value_float")) { auto t = make(DT_F32, {}); t->f[0] = n.af("value_float", 0); set_const(t); continue; }
            if (n.has("value_int")) { auto t = make(DT_I64, {}); t->i[0] = n.ai("value_int", 0); set_const(t); continue; }
            if (n.has("value_ints")) { auto v = n.aints("value_ints"); auto t = make(DT_I64, {(int64_t)v.size()}); t->i = v; set_const(t); continue; }
            if (n.has("value_floats")) { auto& v = n.attr.at("value_floats").floats; auto t = make(DT_F32, {(int64_t)v.size()}); t->f = v; set_const(t); continue; }
            throw std::runtime_error("fold: Constant without a supported value attribute");
        }

        if (all_const) {
            // ---------------------------------------------------------------- constant evaluation
            if (op == "Add" || op == "Sub" || op == "Mul" || op == "Div" || op == "Pow" || op == "Equal" || op == "Less" || op == "Greater" ||
                op == "LessOrEqual" || op == "GreaterOrEqual" || op == "And" || op == "Or" || op == "Xor" || op == "Max" || op == "Min" || op == "Mod") {
                set_const(binary(op, C(0), C(1))); continue;
            }
            if (op == "Not") { auto o = make(DT_BOOL, C(0).shape); for (size_t k = 0; k < o->i.size(); ++k) o->i[k] = !C(0).i[k]; set_const(o); continue; }
            if (op == "Neg" || op == "Sqrt" || op == "Floor" || op == "Ceil" || op == "Abs" || op == "Exp" || op == "Reciprocal") {
                const HTensor& a = C(0); auto o = make(a.dtype, a.shape);
                for (int64_t k = 0; k < a.numel(); ++k) {
                    if (a.is_float()) { float x = a.f[k]; o->f[k] = op == "Neg" ? -x : op == "Sqrt" ? std::sqrt(x) : op == "Floor" ? std::floor(x) : op == "Ceil" ? std::ceil(x) : op == "Abs" ? std::fabs(x) : op == "Exp" ? std::exp(x) : 1.f / x; }
                    else { int64_t x = a.i[k]; o->i[k] = op == "Neg" ? -x : op == "Abs" ? std::llabs(x) : x; }
                }
                set_const(o); continue;
            }
            if (op == "Identity" || op == "Dropout") { set_const(in[0]->c); continue; }
            if (op == "Cast") { set_const(cast_to(C(0), (int)n.ai("to", DT_F32))); continue; }
            if (op == "Where") {
                const HTensor &c = C(0), &a = C(1), &b = C(2);
                Shape os = broadcast_shape(broadcast_shape(c.shape, a.shape), b.shape);
                auto o = make(a.is_float() || b.is_float() ? DT_F32 : a.dtype, os);
                Bcast ic(os, c.shape), ia(os, a.shape), ib(os, b.shape);
                for (int64_t k = 0; k < prod(os); ++k) {
                    bool s = c.i[ic(k)] != 0;
                    if (o->is_float()) o->f[k] = (float)(s ? getd(a, ia(k)) : getd(b, ib(k)));
                    else o->i[k] = s ? a.i[ia(k)] : b.i[ib(k)];
                }
                set_const(o); continue;
            }
            if (op == "Unsqueeze") { auto o = std::make_shared<HTensor>(C(0)); o->shape = unsqueeze_shape(C(0).shape, axes_arg(n, in, 1)); set_const(o); continue; }
            if (op == "Squeeze") { auto ax = axes_arg(n, in, 1); auto o = std::make_shared<HTensor>(C(0)); o->shape = squeeze_shape(C(0).shape, ax, ax.empty()); set_const(o); continue; }
            if (op == "Reshape") { auto o = std::make_shared<HTensor>(C(0)); o->shape = reshape_target(C(0).shape, ints_of(C(1)), n.ai("allowzero", 0) != 0); set_const(o); continue; }
            if (op == "Flatten") { auto o = std::make_shared<HTensor>(C(0)); int64_t ax = norm_axis(n.ai("axis", 1), C(0).rank() + 1); int64_t a = 1; for (int64_t k = 0; k < ax; ++k) a *= C(0).shape[k]; o->shape = {a, a ? C(0).numel() / a : 0}; set_const(o); continue; }
            if (op == "Concat") {
                int r = C(0).rank(); int64_t ax = norm_axis(n.ai("axis", 0), r);
                Shape os = C(0).shape; os[ax] = 0;
                bool fl = false;
                for (size_t k = 0; k < in.size(); ++k) { os[ax] += C(k).shape.empty() ? 1 : C(k).shape[ax]; fl |= C(k).is_float(); }
                auto o = make(fl ? DT_F32 : C(0).dtype, os);
                int64_t outer = 1, inner = 1;
                for (int64_t k = 0; k < ax; ++k) outer *= os[k];
                for (int k = (int)ax + 1; k < r; ++k) inner *= os[k];
                int64_t off = 0;
                for (size_t t = 0; t < in.size(); ++t) {
                    const HTensor& a = C(t); int64_t da = a.shape[ax];
                    for (int64_t u = 0; u < outer; ++u) for (int64_t v = 0; v < da * inner; ++v) {
                        int64_t dst = (u * os[ax] + off) * inner + v, src = u * da * inner + v;
                        if (fl) o->f[dst] = (float)getd(a, src); else o->i[dst] = a.i[src];
                    }
                    off += da;
                }
                set_const(o); continue;
            }
            if (op == "Slice") {
                std::vector<int64_t> starts, ends, axes, steps;
                if (n.has("starts")) { starts = n.aints("starts"); ends = n.aints("ends"); axes = n.aints("axes"); }
                else { starts = ints_of(C(1)); ends = ints_of(C(2)); if (has_in(3)) axes = ints_of(C(3)); if (has_in(4)) steps = ints_of(C(4)); }
                const HTensor& a = C(0);
                SliceSpec sp = slice_spec(a.shape, starts, ends, axes, steps);
                Shape ist = strides_of(a.shape), ost = strides_of(sp.oshape);
                set_const(gather_elems(a, sp.oshape, [&](int64_t flat) {
                    int64_t s = 0;
                    for (size_t k = 0; k < sp.oshape.size(); ++k) { int64_t c = (flat / ost[k]) % sp.oshape[k]; s += (sp.start[k] + c * sp.step[k]) * ist[k]; }
                    return s; }));
                continue;
            }
            if (op == "Gather") {
                const HTensor &a = C(0), &idx = C(1);
                int r = a.rank(); int64_t ax = norm_axis(n.ai("axis", 0), r);
                Shape os(a.shape.begin(), a.shape.begin() + ax);
                os.insert(os.end(), idx.shape.begin(), idx.shape.end());
                os.insert(os.end(), a.shape.begin() + ax + 1, a.shape.end());
                int64_t inner = 1; for (int k = (int)ax + 1; k < r; ++k) inner *= a.shape[k];
                int64_t ni = idx.numel();
                set_const(gather_elems(a, os, [&](int64_t flat) {
                    int64_t in_i = flat % inner, j = (flat / inner) % std::max<int64_t>(ni, 1), outer = flat / inner / std::max<int64_t>(ni, 1);
                    int64_t ix = idx.i[j]; if (ix < 0) ix += a.shape[ax];
                    if (ix < 0 || ix >= a.shape[ax]) throw std::runtime_error("fold: gather index out of range");
                    return (outer * a.shape[ax] + ix) * inner + in_i; }));
                continue;
            }
            if (op == "ConstantOfShape") {
                Shape os = ints_of(C(0));
                auto it = n.attr.find("value");
                int dt = DT_F32; double v = 0;
                if (it != n.attr.end() && it->second.t) { dt = it->second.t->dtype; v = getd(*it->second.t, 0); }
                auto o = make(dt, os);
                if (o->is_float()) std::fill(o->f.begin(), o->f.end(), (float)v); else std::fill(o->i.begin(), o->i.end(), (int64_t)v);
                set_const(o); continue;
            }
            if (op == "Expand") {
                const HTensor& a = C(0); Shape os = broadcast_shape(a.shape, ints_of(C(1)));
                Bcast ia(os, a.shape);
                set_const(gather_elems(a, os, [&](int64_t k) { return ia(k); })); continue;
            }
            if (op == "Range") {
                const HTensor &s = C(0), &e = C(1), &d = C(2);
                if (s.is_float()) {
                    int64_t cnt = std::max<int64_t>(0, (int64_t)std::ceil((e.f[0] - s.f[0]) / d.f[0]));
                    auto o = make(DT_F32, {cnt}); for (int64_t k = 0; k < cnt; ++k) o->f[k] = s.f[0] + k * d.f[0]; set_const(o);
                } else {
                    if (d.i[0] == 0) throw std::runtime_error("fold: Range with zero delta");
                    int64_t cnt = std::max<int64_t>(0, (int64_t)std::ceil((double)(e.i[0] - s.i[0]) / (double)d.i[0]));
                    auto o = make(s.dtype, {cnt}); for (int64_t k = 0; k < cnt; ++k) o->i[k] = s.i[0] + k * d.i[0]; set_const(o);
                }
                continue;
            }
            if (op == "Transpose") {
                const HTensor& a = C(0); int r = a.rank();
                auto perm = n.aints("perm"); if (perm.empty()) for (int k = r - 1; k >= 0; --k) perm.push_back(k);
                check_perm(perm, r);
                Shape os(r); for (int k = 0; k < r; ++k) os[k] = a.shape[perm[k]];
                Shape ist = strides_of(a.shape), ost = strides_of(os);
                set_const(gather_elems(a, os, [&](int64_t flat) { int64_t s = 0; for (int k = 0; k < r; ++k) s += ((flat / ost[k]) % os[k]) * ist[perm[k]]; return s; }));
                continue;
            }
            if (op == "ScatterND") {
                const HTensor &a = C(0), &idx = C(1), &upd = C(2);
                auto o = std::make_shared<HTensor>(a);
                int64_t kdim = idx.shape.back(), cnt = idx.numel() / std::max<int64_t>(kdim, 1);
                Shape st = strides_of(a.shape);
                int64_t slice = 1; for (int k = (int)kdim; k < a.rank(); ++k) slice *= a.shape[k];
                for (int64_t c = 0; c < cnt; ++c) {
                    int64_t base = 0;
                    for (int64_t k = 0; k < kdim; ++k) { int64_t ix = idx.i[c * kdim + k]; if (ix < 0) ix += a.shape[k]; if (ix < 0 || ix >= a.shape[k]) throw std::runtime_error("fold: ScatterND index out of range"); base += ix * st[k]; }
                    for (int64_t e = 0; e < slice; ++e) { if (a.is_float()) o->f[base + e] = (float)getd(upd, c * slice + e); else o->i[base + e] = upd.is_float() ? (int64_t)upd.f[c * slice + e] : upd.i[c * slice + e]; }
                }
                set_const(o); continue;
            }
            if (op == "ReduceProd" || op == "ReduceSum" || op == "ReduceMax" || op == "ReduceMin") {
                const HTensor& a = C(0);
                if (a.rank() > 1) throw std::runtime_error("fold: " + op + " only on vectors");
                auto o = make(a.dtype, n.ai("keepdims", 1) ? Shape{1} : Shape{});
                double acc = op == "ReduceProd" ? 1 : op == "ReduceSum" ? 0 : getd(a, 0);
                for (int64_t k = 0; k < a.numel(); ++k) { double x = getd(a, k); acc = op == "ReduceProd" ? acc * x : op == "ReduceSum" ? acc + x : op == "ReduceMax" ? std::max(acc, x) : std::min(acc, x); }
                if (o->is_float()) o->f[0] = (float)acc; else o->i[0] = (int64_t)acc;
                set_const(o); continue;
            }
            if (op == "Size") { auto o = make(DT_I64, {}); o->i[0] = C(0).numel(); set_const(o); continue; }
            throw std::runtime_error("fold: cannot constant-fold op " + op);
        }

        // -------------------------------------------------------------------- runtime node: shape inference
        const Shape& s0 = in[0] ? in[0]->shape : Shape{};
        int dt0 = in[0] ? in[0]->dtype : DT_F32;
        if (op == "Conv" || op == "ConvTranspose") {
            const Shape& w = in[1]->shape;
            if (s0.size() != 4 || w.size() != 4) throw std::runtime_error("graph: only 2-D convolutions are supported");
            auto strides = n.aints("strides"); if (strides.empty()) strides = {1, 1};
            auto pads = n.aints("pads"); if (pads.empty()) pads = {0, 0, 0, 0};
            auto dil = n.aints("dilations"); if (dil.empty()) dil = {1, 1};
            if (op == "Conv") {
                int64_t oh = (s0[2] + pads[0] + pads[2] - dil[0] * (w[2] - 1) - 1) / strides[0] + 1;
                int64_t ow = (s0[3] + pads[1] + pads[3] - dil[1] * (w[3] - 1) - 1) / strides[1] + 1;
                set_dyn({s0[0], w[0], oh, ow}, dt0);
            } else {
                auto op_ = n.aints("output_padding"); if (op_.empty()) op_ = {0, 0};
                int64_t oh = (s0[2] - 1) * strides[0] - pads[0] - pads[2] + dil[0] * (w[2] - 1) + op_[0] + 1;
                int64_t ow = (s0[3] - 1) * strides[1] - pads[1] - pads[3] + dil[1] * (w[3] - 1) + op_[1] + 1;
                set_dyn({s0[0], w[1] * n.ai("group", 1), oh, ow}, dt0);
            }
        } else if (op == "LeakyRelu" || op == "Relu" || op == "Sigmoid" || op == "Erf" || op == "Tanh" || op == "Sqrt" || op == "Exp" || op == "Neg" ||
                   op == "Clip" || op == "Softmax" || op == "LayerNormalization" || op == "Identity" || op == "Dropout" || op == "Abs" || op == "Reciprocal" || op == "Gelu") {
            set_dyn(s0, dt0);
        } else if (op == "Cast") {
            set_dyn(s0, (int)n.ai("to", DT_F32));
        } else if (op == "Add" || op == "Sub" || op == "Mul" || op == "Div" || op == "Pow" || op == "Max" || op == "Min") {
            set_dyn(broadcast_shape(in[0]->shape, in[1]->shape), in[0]->is_const ? in[1]->dtype : dt0);
        } else if (op == "MatMul") {
            Shape a = in[0]->shape, b = in[1]->shape;
            if (a.size() < 2 || b.size() < 2 || a.back() != b[b.size() - 2]) throw std::runtime_error("graph: MatMul shape mismatch");
            Shape ba(a.begin(), a.end() - 2), bb(b.begin(), b.end() - 2);
            Shape o = broadcast_shape(ba, bb);
            o.push_back(a[a.size() - 2]); o.push_back(b.back());
            set_dyn(o, dt0);
        } else if (op == "Gemm") {
            Shape a = in[0]->shape, b = in[1]->shape;
            int64_t M = n.ai("transA", 0) ? a[1] : a[0], N = n.ai("transB", 0) ? b[0] : b[1];
            set_dyn({M, N}, dt0);
        } else if (op == "Reshape") {
            set_dyn(reshape_target(s0, ints_of(C(1)), n.ai("allowzero", 0) != 0), dt0);
        } else if (op == "Flatten") {
            int64_t ax = norm_axis(n.ai("axis", 1), (int)s0.size() + 1); int64_t a = 1; for (int64_t k = 0; k < ax; ++k) a *= s0[k];
            set_dyn({a, prod(s0) / std::max<int64_t>(a, 1)}, dt0);
        } else if (op == "Transpose") {
            auto perm = n.aints("perm"); int r = (int)s0.size(); if (perm.empty()) for (int k = r - 1; k >= 0; --k) perm.push_back(k);
            check_perm(perm, r);
            Shape o(r); for (int k = 0; k < r; ++k) o[k] = s0[perm[k]];
            set_dyn(o, dt0);
        } else if (op == "Unsqueeze") {
            set_dyn(unsqueeze_shape(s0, axes_arg(n, in, 1)), dt0);
        } else if (op == "Squeeze") {
            auto ax = axes_arg(n, in, 1); set_dyn(squeeze_shape(s0, ax, ax.empty()), dt0);
        } else if (op == "Slice") {
            std::vector<int64_t> starts, ends, axes, steps;
            if (n.has("starts")) { starts = n.aints("starts"); ends = n.aints("ends"); axes = n.aints("axes"); }
            else { starts = ints_of(C(1)); ends = ints_of(C(2)); if (has_in(3)) axes = ints_of(C(3)); if (has_in(4)) steps = ints_of(C(4)); }
            set_dyn(slice_spec(s0, starts, ends, axes, steps).oshape, dt0);
        } else if (op == "Split") {        // several outputs (qkv.unbind(0) exports as Split + Squeeze): sizes from the attribute, the second input, or equal parts
            int64_t ax = norm_axis(n.ai("axis", 0), (int)s0.size());
            std::vector<int64_t> sizes = n.aints("split");
            if (sizes.empty() && has_in(1)) sizes = ints_of(C(1));
            if (sizes.empty()) { const int64_t k = (int64_t)n.out.size(); if (k <= 0 || s0[ax] % k) throw std::runtime_error("graph: Split \"" + n.name + "\" does not divide its axis evenly"); sizes.assign((size_t)k, s0[ax] / k); }
            int64_t total = 0; for (int64_t v : sizes) { if (v < 0) throw std::runtime_error("graph: Split with a negative size"); total += v; }
            if (sizes.size() != n.out.size() || total != s0[ax]) throw std::runtime_error("graph: Split \"" + n.name + "\" sizes do not match its axis");
            for (size_t k = 0; k < sizes.size(); ++k) { Value v; v.shape = s0; v.shape[ax] = sizes[k]; v.dtype = dt0; g.vals[n.out[k]] = v; }
        } else if (op == "Concat") {
            int64_t ax = norm_axis(n.ai("axis", 0), (int)s0.size()); Shape o = s0; o[ax] = 0;
            for (auto* v : in) o[ax] += v->shape[ax];
            set_dyn(o, dt0);
        } else if (op == "Gather") {
            const HTensor& idx = C(1); int64_t ax = norm_axis(n.ai("axis", 0), (int)s0.size());
            Shape o(s0.begin(), s0.begin() + ax); o.insert(o.end(), idx.shape.begin(), idx.shape.end()); o.insert(o.end(), s0.begin() + ax + 1, s0.end());
            set_dyn(o, dt0);
        } else if (op == "Pad") {
            std::vector<int64_t> pads = n.has("pads") ? n.aints("pads") : ints_of(C(1));
            size_t r = s0.size(); if (pads.size() != 2 * r) throw std::runtime_error("graph: Pad pads size mismatch");
            Shape o(r); for (size_t k = 0; k < r; ++k) o[k] = s0[k] + pads[k] + pads[k + r];
            set_dyn(o, dt0);
        } else if (op == "DepthToSpace") {
            int64_t b = n.ai("blocksize", 1);
            set_dyn({s0[0], s0[1] / (b * b), s0[2] * b, s0[3] * b}, dt0);
        } else if (op == "GlobalAveragePool") {
            set_dyn({s0[0], s0[1], 1, 1}, dt0);
        } else if (op == "ReduceMean") {
            auto axes = axes_arg(n, in, 1); bool keep = n.ai("keepdims", 1) != 0; int r = (int)s0.size();
            for (auto& a : axes) a = norm_axis(a, r);
            Shape o; for (int k = 0; k < r; ++k) { bool red = axes.empty() || std::find(axes.begin(), axes.end(), k) != axes.end(); if (!red) o.push_back(s0[k]); else if (keep) o.push_back(1); }
            set_dyn(o, dt0);
        } else if (op == "Expand") {
            set_dyn(broadcast_shape(s0, ints_of(C(1))), dt0);
        } else {
            throw std::runtime_error("graph: unsupported operator " + op);
        }
        g.nodes.push_back(&n);
        for (auto& o : n.out) g.producer[o] = &n;
        for (auto& i : n.in) if (!i.empty() && !g.vals[i].is_const) g.consumers[i].push_back(&n);
    }
    if (!g.vals.count(g.output)) throw std::runtime_error("graph: output tensor is never produced");
    if (g.vals[g.output].shape.size() != 4)
        throw std::runtime_error("model has invalid IO tensor shape: expected 4 dims, got " + std::to_string(g.vals[g.output].shape.size()));
    simplify_graph(g);
    return g;
}

}  // namespace w2x
