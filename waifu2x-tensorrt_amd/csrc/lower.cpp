#include "lower.h"

#include <algorithm>
#include <cmath>
#include <list>
#include <set>
#include <sstream>
#include <stdexcept>

#include "common.h"
#include "support.h"
#include "switches.h"

namespace w2x {
namespace {

using Shape = std::vector<int64_t>;

struct LVal {
    enum Kind { MAP, WINROWS } kind = MAP;
    View v;                // base view (MAP) / window-order rows tensor (WINROWS: v.t, uncropped)
    int C = 0;             // logical channels
    bool nchw = false;     // ONNX-logical layout of the value (storage is always channel-last)
    bool ln = false;       // lazy LayerNorm over C pending on this value
    const HTensor* gamma = nullptr;
    const HTensor* beta = nullptr;
    float eps = 1e-5f;
    int roll_y = 0, roll_x = 0;  // value[y][x] = base[(y+roll_y)%H][(x+roll_x)%W]
    int ws = 0;            // WINROWS: window size
    int table = -1;        // WINROWS: blob with window-order -> pixel table this value was gathered with
    int H = 0, W = 0;      // WINROWS: token map extent
};

std::string shape_str(const Shape& s) {
    std::ostringstream o; o << "[";
    for (size_t k = 0; k < s.size(); ++k) o << (k ? "," : "") << s[k];
    o << "]"; return o.str();
}

struct Lowerer {
    const FoldedGraph& g;
    Plan plan;
    std::map<std::string, LVal> vals;
    std::set<const Node*> done;
    std::map<int, int> tensor_producer;  // tensor -> op index that wrote it (for stats_out / pool_out requests)
    std::map<std::string, int> blob_cache;
    std::map<int, int> pending_gate;     // map tensor -> fp32 [B][C] squeeze-excite gate not yet applied to it (folded into its consumers)

    explicit Lowerer(const FoldedGraph& g_) : g(g_) {}

    [[noreturn]] void fail(const Node* n, const std::string& why) {
        std::string s = "cannot lower node " + (n ? n->op + " \"" + n->name + "\"" : std::string("<graph>")) + ": " + why;
        throw std::runtime_error(s);
    }

    // ---- graph helpers
    const Node* only_user(const std::string& name) {
        if (name == g.output) return nullptr;
        auto it = g.consumers.find(name);
        if (it == g.consumers.end() || it->second.size() != 1) return nullptr;
        return it->second[0];
    }
    std::vector<const Node*> users(const std::string& name) {
        auto it = g.consumers.find(name);
        return it == g.consumers.end() ? std::vector<const Node*>{} : it->second;
    }
    const Shape& shp(const std::string& name) { return g.val(name).shape; }
    bool is_c(const std::string& name) { return !name.empty() && g.is_const(name); }
    // the runtime operand / constant operand of a binary node
    bool split_binary(const Node* n, std::string& dyn, const HTensor*& c) {
        if (n->in.size() != 2) return false;
        if (is_c(n->in[0]) && !is_c(n->in[1])) { dyn = n->in[1]; c = &g.cst(n->in[0]); return true; }
        if (is_c(n->in[1]) && !is_c(n->in[0])) { dyn = n->in[0]; c = &g.cst(n->in[1]); return true; }
        return false;
    }
    static bool scalar_near(const HTensor* c, double v, double tol = 1e-4) {
        return c && c->numel() == 1 && c->is_float() && std::fabs(c->f[0] - v) <= tol * std::max(1.0, std::fabs(v));
    }
    static std::vector<int64_t> perm_of(const Node* n) { return n->aints("perm"); }

    // ---- plan helpers
    int new_tensor(int B, int H, int W, int C, int elt = 0) {   // elt 0: an activation map in the plan's precision
        TensorDesc t; t.B = B; t.H = H; t.W = W; t.C = C; t.elt = elt ? elt : plan.elt;
        plan.tensors.push_back(t);
        return (int)plan.tensors.size() - 1;
    }
    int add_blob(std::vector<uint8_t>&& bytes) {
        std::string key((const char*)bytes.data(), bytes.size());
        if (bytes.size() < (1u << 22)) {  // dedupe small tables (window tables, masks)
            auto it = blob_cache.find(key);
            if (it != blob_cache.end()) return it->second;
        }
        Blob b; b.data = std::move(bytes);
        plan.blobs.push_back(std::move(b));
        int id = (int)plan.blobs.size() - 1;
        if (key.size() < (1u << 22)) blob_cache[key] = id;
        return id;
    }
    int blob_f32(const std::vector<float>& v) { std::vector<uint8_t> b(v.size() * 4); memcpy(b.data(), v.data(), b.size()); return add_blob(std::move(b)); }
    int blob_i32(const std::vector<int32_t>& v) { std::vector<uint8_t> b(v.size() * 4); memcpy(b.data(), v.data(), b.size()); return add_blob(std::move(b)); }
    int blob_w(const std::vector<float>& v) { return plan.elt == 4 ? blob_f32(v) : blob_f16(v); }   // weights / bias tables in the plan's precision
    int blob_f16(const std::vector<float>& v) {
        std::vector<uint8_t> b(v.size() * 2);
        for (size_t k = 0; k < v.size(); ++k) { uint16_t h = f32_to_f16(v[k]); memcpy(&b[2 * k], &h, 2); }
        return add_blob(std::move(b));
    }
    static int stored_c(int c) { return c <= 4 ? 4 : round_up(c, 8); }

    int window_table(int H, int W, int ws, int ry, int rx, bool inverse_roll) {
        // forward (gather): value[y][x] = base[(y+ry)%H][(x+rx)%W], partitioned into ws x ws windows, row-major windows and tokens.
        // inverse_roll (scatter): u[y'][x'] (un-partitioned) lands at out[(y'-ry) mod H][(x'-rx) mod W].
        std::vector<int32_t> t((size_t)H * W);
        int nwx = W / ws;
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            int wy = y / ws, ty = y % ws, wx = x / ws, tx = x % ws;
            int row = ((wy * nwx + wx) * ws + ty) * ws + tx;
            int sy, sx;
            if (!inverse_roll) { sy = (y + ry) % H; sx = (x + rx) % W; }
            else { sy = ((y - ry) % H + H) % H; sx = ((x - rx) % W + W) % W; }
            t[row] = sy * W + sx;
        }
        return blob_i32(t);
    }

    void request_stats(const Node* n, const LVal& v) {
        // LayerNorm statistics of the rows of tensor v.v.t are produced by the epilogue of the op that wrote it.
        auto it = tensor_producer.find(v.v.t);
        if (it == tensor_producer.end()) fail(n, "LayerNorm input is not produced by a fused op");
        Op& p = plan.ops[it->second];
        if (p.kind != OP_GEMM) fail(n, "LayerNorm input producer cannot emit row statistics");
        const TensorDesc& td = plan.tensors[v.v.t];
        if (v.v.y0 || v.v.x0 || v.v.H != td.H || v.v.W != td.W) fail(n, "LayerNorm over a cropped view is not supported");
        if (p.g.Cout != v.C) fail(n, "LayerNorm width differs from producer row width");
        // the statistics of a row come out of ONE workgroup tile of the producing kernel (k_gemm.hip / k_f32.hip: tiles of up to 192 columns);
        // wider token rows would need a separate reduction
        if (p.g.omode != O_PIXSHUF && !gemm_row_stats_supported(p.g.N))
            fail(n, "LayerNorm over rows of " + std::to_string(p.g.N) + " channels: the producing kernel emits row statistics for widths 32, 48, 64, 96, 128 and 192 only");
        if (p.g.stats_out < 0) { p.g.stats_out = new_tensor(td.B, td.H, td.W, 2, 4); p.g.ln_eps = v.eps; }
        else if (p.g.ln_eps != v.eps) fail(n, "two LayerNorms with different eps on one tensor");
    }

    // ---- roll detection: Concat(axis)[Slice(x, s:end), Slice(x, 0:s)]
    bool try_roll_axis(const std::string& name, int& axis, int& shift, std::string& out, std::vector<const Node*>& used) {
        auto us = users(name);
        if (us.size() != 2 || us[0]->op != "Slice" || us[1]->op != "Slice") return false;
        const Node* cat = only_user(us[0]->out[0]);
        if (!cat || cat->op != "Concat" || only_user(us[1]->out[0]) != cat || cat->in.size() != 2) return false;
        auto params = [&](const Node* s, int64_t& st, int64_t& en, int64_t& ax) {
            if (s->in.size() < 4 || !is_c(s->in[1]) || !is_c(s->in[2]) || !is_c(s->in[3])) return false;
            if (s->in.size() > 4 && !s->in[4].empty()) { if (!is_c(s->in[4]) || g.cst(s->in[4]).i[0] != 1) return false; }
            st = g.cst(s->in[1]).i[0]; en = g.cst(s->in[2]).i[0]; ax = g.cst(s->in[3]).i[0]; return g.cst(s->in[1]).numel() == 1;
        };
        const Node* first = g.producer.at(cat->in[0]); const Node* second = g.producer.at(cat->in[1]);
        int64_t s1, e1, a1, s2, e2, a2;
        if (!params(first, s1, e1, a1) || !params(second, s2, e2, a2) || a1 != a2 || a1 != cat->ai("axis", -99)) return false;
        int64_t D = shp(name)[a1];
        if (s1 < 0) s1 += D; if (e2 < 0) e2 += D;
        if (e1 < D || s2 != 0 || e2 != s1) return false;
        axis = (int)a1; shift = (int)s1; out = cat->out[0];
        used.push_back(us[0]); used.push_back(us[1]); used.push_back(cat);
        return true;
    }

    // ---------------------------------------------------------------------------------------------------------
    // GEMM-like emission with epilogue look-ahead
    struct Pending {
        GemmOp op;
        std::string cur;       // ONNX tensor currently representing the op's value
        bool nchw = false;     // layout of the ONNX value at `cur`
        bool d2s_dcr = false;  // O_PIXSHUF: the file's DepthToSpace is in DCR mode (ONNX column (dy r + dx) C' + c) instead of CRD (c r r + dy r + dx)
        int outH = 0, outW = 0, outC = 0;  // logical output geometry (before pixshuf)
        LVal::Kind okind = LVal::MAP;
        int ws = 0, H = 0, W = 0;          // WINROWS bookkeeping
        std::vector<float> w;  // [N][K] row-major fp32 before packing
        std::vector<float> bias;
        double flops = 0;
        std::string name;
    };

    // absorb `Add(const)` following cur as bias (MatMul style)
    void absorb_bias(Pending& p) {
        const Node* u = only_user(p.cur);
        std::string dyn; const HTensor* c = nullptr;
        if (!u || u->op != "Add" || !split_binary(u, dyn, c) || dyn != p.cur) return;
        if (!c->is_float() || (c->numel() != p.op.N)) return;
        if (c->shape.size() > 1) for (size_t k = 0; k + 1 < c->shape.size(); ++k) if (c->shape[k] != 1) return;
        for (int n = 0; n < p.op.N; ++n) p.bias[n] += c->f[n];
        done.insert(u); p.cur = u->out[0];
    }

    void absorb_act(Pending& p) {
        const Node* u = only_user(p.cur);
        if (u && u->op == "LeakyRelu" && p.op.act == ACT_NONE) { p.op.act = ACT_LEAKY; p.op.alpha = u->af("alpha", 0.01f); done.insert(u); p.cur = u->out[0]; return; }
        if (u && u->op == "Relu" && p.op.act == ACT_NONE) { p.op.act = ACT_RELU; done.insert(u); p.cur = u->out[0]; return; }
        if (u && u->op == "Sigmoid" && p.op.act == ACT_NONE) { p.op.act = ACT_SIGMOID; done.insert(u); p.cur = u->out[0]; return; }
        // opset 20 has the operator itself: Gelu(approximate = "none") is the erf form the kernels compute; the tanh approximation is another function
        if (u && u->op == "Gelu" && p.op.act == ACT_NONE) {
            auto ap = u->attr.find("approximate");
            if (ap != u->attr.end() && ap->second.s != "none") fail(u, "only Gelu approximate=none (the erf form) is supported, the file asks for \"" + ap->second.s + "\"");
            p.op.act = ACT_GELU; done.insert(u); p.cur = u->out[0]; return;
        }
        // GELU(erf): t -> {Div(sqrt2) | Mul(1/sqrt2)} -> Erf -> Add(1) -> Mul(t, .) -> Mul(0.5)
        auto us = users(p.cur);
        if (us.size() == 2 && p.op.act == ACT_NONE && p.cur != g.output) {
            const Node *d = nullptr, *m = nullptr;
            for (auto* x : us) { if (x->op == "Div" || (x->op == "Mul" && (is_c(x->in[0]) || is_c(x->in[1])))) d = x; else if (x->op == "Mul") m = x; }
            if (!d || !m) return;
            std::string dyn; const HTensor* c = nullptr;
            if (!split_binary(d, dyn, c) || dyn != p.cur) return;
            if (!(d->op == "Div" ? scalar_near(c, std::sqrt(2.0)) : scalar_near(c, 1.0 / std::sqrt(2.0)))) return;
            const Node* e = only_user(d->out[0]); if (!e || e->op != "Erf") return;
            const Node* a = only_user(e->out[0]); if (!a || a->op != "Add" || !split_binary(a, dyn, c) || !scalar_near(c, 1.0)) return;
            if (only_user(a->out[0]) != m) return;
            if (!((m->in[0] == p.cur && m->in[1] == a->out[0]) || (m->in[1] == p.cur && m->in[0] == a->out[0]))) return;
            const Node* h = only_user(m->out[0]); if (!h || h->op != "Mul" || !split_binary(h, dyn, c) || !scalar_near(c, 0.5)) return;
            for (auto* x : {d, e, a, m, h}) done.insert(x);
            p.op.act = ACT_GELU; p.cur = h->out[0];
        }
    }

    // pixel shuffle: (nhwc) Transpose(0,3,1,2) -> DepthToSpace(CRD) [-> Transpose(0,2,3,1)]
    void absorb_pixshuf(Pending& p) {
        if (p.nchw || p.okind != LVal::MAP || p.op.omode != O_ROWS) return;
        const Node* t = only_user(p.cur);
        if (!t || t->op != "Transpose" || perm_of(t) != std::vector<int64_t>{0, 3, 1, 2}) return;
        const Node* d = only_user(t->out[0]);
        if (!d || d->op != "DepthToSpace") {
            // plain layout change
            done.insert(t); p.cur = t->out[0]; p.nchw = true; return;
        }
        // CRD is what torch's pixel_shuffle exports to; DCR is the operator's default (mode absent) and what other exporters write: the two differ in which ONNX column
        // holds sub-pixel (dy, dx) of channel c - the weight rows are re-ordered accordingly in finish(), the kernels see (dy, dx, c) either way
        auto mode = d->attr.find("mode");
        if (mode != d->attr.end() && mode->second.s != "CRD" && mode->second.s != "DCR") fail(d, "DepthToSpace mode \"" + mode->second.s + "\" (CRD or DCR expected)");
        p.d2s_dcr = mode == d->attr.end() || mode->second.s == "DCR";
        int r = (int)d->ai("blocksize", 1);
        if (p.op.N % (r * r)) fail(d, "channel count not divisible by blocksize^2");
        done.insert(t); done.insert(d);
        p.op.omode = O_PIXSHUF; p.op.r = r; p.outC = p.op.N / (r * r);
        p.cur = d->out[0]; p.nchw = true;
        const Node* t2 = only_user(p.cur);
        if (t2 && t2->op == "Transpose" && perm_of(t2) == std::vector<int64_t>{0, 2, 3, 1}) { done.insert(t2); p.cur = t2->out[0]; p.nchw = false; }
    }

    // window un-partition (+ reverse roll): Reshape[B,H/ws,W/ws,ws,ws,C] -> Transpose(0,1,3,2,4,5) -> Reshape[B,H,W,C] [-> roll]
    void absorb_unpartition(Pending& p) {
        if (p.okind != LVal::WINROWS) return;
        const Node* r1 = only_user(p.cur);
        if (!r1 || r1->op != "Reshape") return;
        const Shape& s1 = shp(r1->out[0]);
        int ws = p.ws, H = p.H, W = p.W;
        if (s1 != Shape{plan.B, H / ws, W / ws, ws, ws, p.op.N}) return;
        const Node* t = only_user(r1->out[0]);
        if (!t || t->op != "Transpose" || perm_of(t) != std::vector<int64_t>{0, 1, 3, 2, 4, 5}) fail(r1, "unexpected window un-partition");
        const Node* r2 = only_user(t->out[0]);
        if (!r2 || r2->op != "Reshape" || shp(r2->out[0]) != Shape{plan.B, H, W, p.op.N}) fail(t, "unexpected window un-partition");
        for (auto* x : {r1, t, r2}) done.insert(x);
        p.cur = r2->out[0];
        int ry = 0, rx = 0;
        for (int pass = 0; pass < 2; ++pass) {
            int ax, sh; std::string out; std::vector<const Node*> used;
            if (!try_roll_axis(p.cur, ax, sh, out, used)) break;
            if (ax == 1) ry = sh; else if (ax == 2) rx = sh; else fail(used[0], "roll on unsupported axis");
            for (auto* x : used) done.insert(x);
            p.cur = out;
        }
        // out[y][x] = u[(y+ry)%H][(x+rx)%W]  <=>  u[y'][x'] lands at out[(y'-ry) mod H][(x'-rx) mod W]
        p.op.omode = O_WIN;
        p.op.win_table = window_table(H, W, ws, ry, rx, true);
        p.okind = LVal::MAP; p.nchw = false; p.outH = H; p.outW = W;
        // torchvision's shifted_window_attention ends with x[:, :H, :W, :] (one Slice per axis): slices that keep the whole [B, H, W, C] map
        const int64_t dims[4] = {plan.B, H, W, p.op.N};
        for (;;) {
            const Node* sl = only_user(p.cur);
            if (!sl || sl->op != "Slice" || sl->in.size() < 3 || !is_c(sl->in[1]) || !is_c(sl->in[2])) break;
            const std::vector<int64_t> starts = g.cst(sl->in[1]).i, ends = g.cst(sl->in[2]).i;
            const std::vector<int64_t> axes = sl->in.size() > 3 && !sl->in[3].empty() ? g.cst(sl->in[3]).i : std::vector<int64_t>{};
            const std::vector<int64_t> steps = sl->in.size() > 4 && !sl->in[4].empty() ? g.cst(sl->in[4]).i : std::vector<int64_t>{};
            bool whole = !starts.empty() && starts.size() == ends.size();
            for (size_t k = 0; k < starts.size() && whole; ++k) {
                int64_t ax = axes.empty() ? (int64_t)k : axes[k]; if (ax < 0) ax += 4;
                if (ax < 0 || ax > 3 || (!steps.empty() && steps[k] != 1)) { whole = false; break; }
                const int64_t D = dims[ax], st = starts[k] < 0 ? starts[k] + D : starts[k], en = ends[k] < 0 ? ends[k] + D : std::min<int64_t>(ends[k], D);
                whole = st == 0 && en == D;
            }
            if (!whole) break;
            done.insert(sl); p.cur = sl->out[0];
        }
    }

    // lower a pure-view producer (crop) ahead of its position in the node list so it can be used as a residual
    void try_materialize_view(const std::string& name) {
        if (vals.count(name)) return;
        auto itp = g.producer.find(name);
        if (itp == g.producer.end()) return;
        const Node* n = itp->second;
        if (done.count(n) || n->op != "Pad") return;
        auto itx = vals.find(n->in[0]);
        if (itx == vals.end()) return;
        lower_pad(n, itx->second);
    }

    void lower_pad(const Node* n, const LVal& x) {
        std::vector<int64_t> pads = n->has("pads") ? n->aints("pads") : g.cst(n->in[1]).i;
        // a Pad that pads nothing is the identity whatever it sits on: torchvision's shifted_window_attention pads every token map to the window
        // multiple (F.pad(x, (0, 0, 0, pad_r, 0, pad_b))), which exports as a Pad node with all-zero pads when the map already is one
        if (std::all_of(pads.begin(), pads.end(), [](int64_t v) { return v == 0; })) { vals[n->out[0]] = x; done.insert(n); return; }
        if (x.kind != LVal::MAP || !x.nchw || x.ln || pads.size() != 8) fail(n, "only spatial crops of NCHW maps are supported");
        if (pads[0] || pads[1] || pads[4] || pads[5]) fail(n, "padding batch/channel dims");
        if (pads[2] > 0 || pads[3] > 0 || pads[6] > 0 || pads[7] > 0) fail(n, "positive padding is not supported (only negative = crop)");
        LVal y = x;
        y.v.y0 -= (int)pads[2]; y.v.x0 -= (int)pads[3];
        y.v.H += (int)(pads[2] + pads[6]); y.v.W += (int)(pads[3] + pads[7]);
        vals[n->out[0]] = y; done.insert(n);
    }

    // a crop right after a convolution is folded into the convolution (only the kept region is computed)
    void absorb_crop(Pending& p) {
        const Node* u = only_user(p.cur);
        if (!u || u->op != "Pad" || p.op.omode != O_ROWS || !p.nchw) return;
        std::vector<int64_t> pads = u->has("pads") ? u->aints("pads") : g.cst(u->in[1]).i;
        if (pads.size() != 8 || pads[0] || pads[1] || pads[4] || pads[5]) return;
        if (pads[2] > 0 || pads[3] > 0 || pads[6] > 0 || pads[7] > 0) return;
        int top = (int)-pads[2], left = (int)-pads[3], bottom = (int)-pads[6], right = (int)-pads[7];
        p.op.a.y0 += top * p.op.stride; p.op.a.x0 += left * p.op.stride;
        p.outH -= top + bottom; p.outW -= left + right;
        p.op.Mrows = p.outH * p.outW; p.op.aW = p.outW;
        done.insert(u); p.cur = u->out[0];
    }

    bool same_geometry(const LVal& v, const Pending& p, int H, int W, int C) {
        return v.kind == LVal::MAP && !v.ln && !v.roll_y && !v.roll_x && v.v.H == H && v.v.W == W && v.C == C && v.nchw == p.nchw;
    }

    void absorb_residuals(Pending& p, int H, int W, int C) {
        for (int k = 0; k < 2; ++k) {
            const Node* u = only_user(p.cur);
            if (!u || u->op != "Add" || u->in.size() != 2) return;
            std::string other = u->in[0] == p.cur ? u->in[1] : u->in[0];
            if (is_c(other)) return;
            try_materialize_view(other);
            auto it = vals.find(other);
            if (it == vals.end() || !same_geometry(it->second, p, H, W, C)) return;
            View& slot = p.op.res.t < 0 ? p.op.res : p.op.res2;
            if (slot.t >= 0) return;
            if (auto pg = pending_gate.find(it->second.v.t); pg != pending_gate.end()) {   // gated skip connection
                // the kernels gate the FIRST residual operand on 8-half pieces (launch_gemm refuses res_scale onto a 4-channel output,
                // the direct 3x3 kernels take no gated operand at all): anything else gets the in-place pass, so that a graph that
                // lowers also runs
                if (&slot == &p.op.res && stored_c(C) % 8 == 0 && !(p.op.amode == A_CONV && p.op.kh == 3)) p.op.res_scale = pg->second;
                else apply_gate(it->second.v.t);
            }
            slot = it->second.v;
            done.insert(u); p.cur = u->out[0];
        }
    }

    void absorb_clip(Pending& p) {
        const Node* u = only_user(p.cur);
        if (!u || u->op != "Clip") return;
        float lo = -INFINITY, hi = INFINITY;
        if (u->has("min")) lo = u->af("min", lo);
        if (u->has("max")) hi = u->af("max", hi);
        if (u->in.size() > 1 && !u->in[1].empty()) lo = g.cst(u->in[1]).f[0];
        if (u->in.size() > 2 && !u->in[2].empty()) hi = g.cst(u->in[2]).f[0];
        p.op.has_clip = 1; p.op.clip_lo = lo; p.op.clip_hi = hi;
        done.insert(u); p.cur = u->out[0];
    }

    // finalize: allocate output, pack weights, register the value
    void finish(Pending& p) {
        GemmOp& o = p.op;
        int B = plan.B;
        int oH = p.outH, oW = p.outW, oC = p.outC;
        if (o.omode == O_PIXSHUF) { oH *= o.r; oW *= o.r; }
        absorb_residuals(p, oH, oW, oC);
        absorb_clip(p);
        int Cs = stored_c(oC);
        o.Cout = oC;
        int ot = new_tensor(B, oH, oW, Cs);
        o.out.t = ot; o.out.H = oH; o.out.W = oW; o.out.y0 = o.out.x0 = 0;
        // pack W^T [Npad][K] fp16.  Column order for pixel shuffle: ONNX column c*r*r + dy*r + dx (CRD) or (dy*r+dx)*C' + c (DCR)  ->  (dy*r+dx)*Cs + c.
        int K = o.K, N = o.N;
        int Np = N;
        std::vector<int> colmap(N);
        if (o.omode == O_PIXSHUF) {
            int rr = o.r * o.r; Np = rr * Cs;
            for (int c = 0; c < oC; ++c) for (int s = 0; s < rr; ++s) colmap[p.d2s_dcr ? s * oC + c : c * rr + s] = s * Cs + c;
        } else {
            Np = Cs;
            for (int n = 0; n < N; ++n) colmap[n] = n;
        }
        const int Kw = round_up(K, 8);   // row stride of the packed weights (zero padded)
        std::vector<float> wt((size_t)Np * Kw, 0.f), bs(Np, 0.f), cs(Np, 0.f);
        for (int n = 0; n < N; ++n) {
            memcpy(&wt[(size_t)colmap[n] * Kw], &p.w[(size_t)n * K], sizeof(float) * K);
            bs[colmap[n]] = p.bias[n];
        }
        if (o.ln) {
            // csum over the fp16-rounded weights so that mean*csum cancels exactly what the MFMA accumulates
            for (int n = 0; n < Np; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += plan.elt == 4 ? wt[(size_t)n * Kw + k] : f16_to_f32(f32_to_f16(wt[(size_t)n * Kw + k])); cs[n] = (float)s; }
            o.csum = blob_f32(cs);
        }
        o.N = Np;
        o.w = blob_w(wt);
        o.bias = blob_f32(bs);
        Op op; op.kind = OP_GEMM; op.g = o; op.flops = p.flops; op.name = p.name;
        plan.ops.push_back(op);
        tensor_producer[ot] = (int)plan.ops.size() - 1;
        LVal v; v.kind = p.okind; v.v = o.out; v.C = oC; v.nchw = p.nchw;
        if (p.okind == LVal::WINROWS) { v.ws = p.ws; v.H = p.H; v.W = p.W; v.table = o.win_table; }
        vals[p.cur] = v;
    }

    // ---- Conv / ConvTranspose
    void lower_conv(const Node* n) {
        const LVal& x = vals.at(n->in[0]);
        if (x.kind != LVal::MAP || !x.nchw || x.ln || x.roll_x || x.roll_y) fail(n, "convolution input must be a plain NCHW map");
        const HTensor& w = g.cst(n->in[1]);
        auto strides = n->aints("strides"); if (strides.empty()) strides = {1, 1};
        auto pads = n->aints("pads"); if (pads.empty()) pads = {0, 0, 0, 0};
        auto dil = n->aints("dilations"); if (dil.empty()) dil = {1, 1};
        if (n->ai("group", 1) != 1 || dil[0] != 1 || dil[1] != 1) fail(n, "grouped/dilated convolution is not supported");
        int Cs_in = plan.tensors[x.v.t].C;
        Pending p; p.name = n->name.empty() ? n->op : n->name;
        p.nchw = true; p.cur = n->out[0];
        GemmOp& o = p.op;
        o.amode = A_CONV; o.a = x.v;
        const Shape& os = shp(n->out[0]);
        if (n->op == "Conv") {
            int Cout = (int)w.shape[0], Cin = (int)w.shape[1], kh = (int)w.shape[2], kw = (int)w.shape[3];
            if (Cin != x.C) fail(n, "channel mismatch");
            if (pads[0] || pads[1] || pads[2] || pads[3]) fail(n, "only valid (pad 0) convolutions are supported");
            if (strides[0] != strides[1] || kh != kw) fail(n, "only square kernels/strides are supported");
            o.kh = kh; o.kw = kw; o.stride = (int)strides[0];
            o.K = kh * kw * Cs_in; o.N = Cout;
            p.outH = (int)os[2]; p.outW = (int)os[3]; p.outC = Cout;
            p.w.assign((size_t)Cout * o.K, 0.f);
            for (int co = 0; co < Cout; ++co) for (int ci = 0; ci < Cin; ++ci) for (int ky = 0; ky < kh; ++ky) for (int kx = 0; kx < kw; ++kx)
                p.w[(size_t)co * o.K + (ky * kw + kx) * Cs_in + ci] = w.f[((size_t)(co * Cin + ci) * kh + ky) * kw + kx];
            p.flops = 2.0 * (double)os[0] * os[1] * os[2] * os[3] * Cin * kh * kw;
        } else {  // ConvTranspose: weight [Cin][Cout][kh][kw]
            int Cin = (int)w.shape[0], Cout = (int)w.shape[1], kh = (int)w.shape[2], kw = (int)w.shape[3];
            if (Cin != x.C) fail(n, "channel mismatch");
            auto opad = n->aints("output_padding");
            if (!opad.empty() && (opad[0] || opad[1])) fail(n, "output_padding is not supported");
            int s = (int)strides[0];
            p.flops = 2.0 * (double)plan.B * Cin * x.v.H * x.v.W * Cout * kh * kw;
            if (kh == 2 && kw == 2 && s == 2 && strides[1] == 2 && !pads[0] && !pads[1] && !pads[2] && !pads[3]) {
                // out[2y+dy][2x+dx][co] = sum_ci x[y][x][ci] W[ci][co][dy][dx]   -> 1x1 "conv" + pixel shuffle(2)
                o.kh = o.kw = 1; o.stride = 1; o.K = Cs_in; o.N = Cout * 4;
                p.outH = x.v.H; p.outW = x.v.W;
                p.w.assign((size_t)o.N * o.K, 0.f);
                for (int co = 0; co < Cout; ++co) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) for (int ci = 0; ci < Cin; ++ci)
                    p.w[(size_t)(co * 4 + dy * 2 + dx) * o.K + ci] = w.f[((size_t)(ci * Cout + co) * 2 + dy) * 2 + dx];
            } else if (kh == 4 && kw == 4 && s == 2 && strides[1] == 2 && pads[0] == 3 && pads[1] == 3 && pads[2] == 3 && pads[3] == 3) {
                // oy = 2*iy - 3 + ky.  Output block u=(oy>>1) reads input rows u..u+2: parity 0 uses (tap 0,ky 3),(tap 1,ky 1);
                // parity 1 uses (tap 1,ky 2),(tap 2,ky 0)  -> valid 3x3 conv producing 2x2 pixel blocks.
                o.kh = o.kw = 3; o.stride = 1; o.K = 9 * Cs_in; o.N = Cout * 4;
                p.outH = x.v.H - 2; p.outW = x.v.W - 2;
                p.w.assign((size_t)o.N * o.K, 0.f);
                auto kmap = [](int parity, int tap) -> int { if (parity == 0) return tap == 0 ? 3 : tap == 1 ? 1 : -1; return tap == 1 ? 2 : tap == 2 ? 0 : -1; };
                for (int co = 0; co < Cout; ++co) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx)
                    for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
                        int ky = kmap(dy, ty), kx = kmap(dx, tx);
                        if (ky < 0 || kx < 0) continue;
                        for (int ci = 0; ci < Cin; ++ci)
                            p.w[(size_t)(co * 4 + dy * 2 + dx) * o.K + (ty * 3 + tx) * Cs_in + ci] = w.f[((size_t)(ci * Cout + co) * 4 + ky) * 4 + kx];
                    }
            } else fail(n, "unsupported transposed convolution geometry");
            o.omode = O_PIXSHUF; o.r = 2;
            p.outC = Cout;
            if (p.outH * 2 != os[2] || p.outW * 2 != os[3]) fail(n, "transposed convolution output size mismatch");
        }
        o.Mrows = p.outH * p.outW; o.aW = p.outW;
        if (auto pg = pending_gate.find(x.v.t); pg != pending_gate.end()) {   // gated input: 1x1 / 2x2 kernels scale it on load
            // (gates ride on 8-half pieces of the operand rows: launch_gemm refuses a_scale on 4-channel or unaligned maps)
            if (o.kh <= 2 && plan.tensors[x.v.t].C % 8 == 0) o.se_scale = pg->second; else apply_gate(x.v.t);
        }
        p.bias.assign(o.N, 0.f);
        if (n->in.size() > 2 && !n->in[2].empty()) {
            const HTensor& b = g.cst(n->in[2]);
            if (n->op == "Conv") for (int k = 0; k < o.N; ++k) p.bias[k] = b.f[k];
            else for (int k = 0; k < o.N; ++k) p.bias[k] = b.f[k / 4];
        }
        done.insert(n);
        absorb_act(p);
        if (n->op == "Conv") absorb_crop(p);
        finish(p);
    }

    // ---- MatMul with constant weight on NHWC maps / window rows
    void lower_matmul(const Node* n) {
        if (!is_c(n->in[1])) fail(n, "MatMul with a runtime right operand outside an attention block");
        const LVal& x = vals.at(n->in[0]);
        const HTensor& w = g.cst(n->in[1]);
        if (w.rank() != 2) fail(n, "weight must be 2-D");
        int K = (int)w.shape[0], N = (int)w.shape[1];
        if (x.nchw) fail(n, "MatMul on an NCHW value");
        if (K != x.C) fail(n, "inner dimension mismatch");
        Pending p; p.name = n->name.empty() ? n->op : n->name;
        p.cur = n->out[0]; p.nchw = false;
        GemmOp& o = p.op;
        const TensorDesc& td = plan.tensors[x.v.t];
        if (td.C != K) fail(n, "MatMul input has padded channels");
        apply_gate(x.v.t);
        o.a = x.v; o.K = K; o.N = N;
        p.w.assign((size_t)N * K, 0.f);
        p.bias.assign(N, 0.f);
        if (x.ln) {
            request_stats(n, x);
            o.ln = 1; o.stats_in = plan.ops[tensor_producer.at(x.v.t)].g.stats_out;
            for (int k = 0; k < K; ++k) for (int c = 0; c < N; ++c) {
                float wv = w.f[(size_t)k * N + c];
                p.w[(size_t)c * K + k] = x.gamma->f[k] * wv;
                if (x.beta) p.bias[c] += x.beta->f[k] * wv;
            }
        } else {
            for (int k = 0; k < K; ++k) for (int c = 0; c < N; ++c) p.w[(size_t)c * K + k] = w.f[(size_t)k * N + c];
        }
        if (x.kind == LVal::WINROWS) {
            // rows already in window order (attention output): plain rows in, window rows out until un-partitioned
            o.amode = A_ROWS; o.Mrows = x.H * x.W; o.aW = x.W;
            p.okind = LVal::WINROWS; p.ws = x.ws; p.H = x.H; p.W = x.W; p.outH = x.H; p.outW = x.W; p.outC = N;
        } else if (x.ws) {
            // window-partitioned view of a map: gather rows through the window table
            o.amode = A_WIN; o.win_table = x.table; o.Mrows = x.v.H * x.v.W; o.aW = x.v.W;
            p.okind = LVal::WINROWS; p.ws = x.ws; p.H = x.v.H; p.W = x.v.W; p.outH = x.v.H; p.outW = x.v.W; p.outC = N;
        } else {
            if (x.roll_x || x.roll_y) fail(n, "MatMul on a rolled map outside a window partition");
            o.amode = A_ROWS; o.Mrows = x.v.H * x.v.W; o.aW = x.v.W;
            p.outH = x.v.H; p.outW = x.v.W; p.outC = N;
        }
        p.flops = 2.0 * (double)plan.B * o.Mrows * K * N;
        done.insert(n);
        absorb_bias(p);
        absorb_act(p);
        absorb_unpartition(p);
        absorb_pixshuf(p);
        finish(p);
    }

    // ---- the attention core on a materialized qkv tensor (window rows)
    void lower_attention(const Node* r5) {
        const LVal& x = vals.at(r5->in[0]);
        if (x.kind != LVal::WINROWS) fail(r5, "attention input is not a window-partitioned tensor");
        const Shape& s5 = shp(r5->out[0]);
        int ws = x.ws, Ntok = ws * ws, nwin = (x.H / ws) * (x.W / ws);
        if (s5.size() != 5 || s5[0] != (int64_t)plan.B * nwin || s5[1] != Ntok || s5[2] != 3) fail(r5, "unexpected qkv reshape " + shape_str(s5));
        int heads = (int)s5[3], hd = (int)s5[4], C = heads * hd;
        if (x.C != 3 * C) fail(r5, "qkv width mismatch");
        const Node* t5 = only_user(r5->out[0]);
        if (!t5 || t5->op != "Transpose" || perm_of(t5) != std::vector<int64_t>{2, 0, 3, 1, 4}) fail(r5, "unexpected qkv permutation");
        // q, k, v out of the [3, B nW, heads, N, d] tensor: three Gathers on axis 0 (qkv[0], qkv[1], qkv[2]) or one Split on axis 0 whose outputs are
        // squeezed (qkv.unbind(0): Split + Squeeze, a Reshape after simplify_graph)
        auto gs = users(t5->out[0]);
        std::string q, k, v;
        if (gs.size() == 1 && gs[0]->op == "Split") {
            const Node* sp = gs[0];
            if (sp->ai("axis", 0) != 0 || sp->out.size() != 3) fail(sp, "unexpected qkv split");
            std::string* dst[3] = {&q, &k, &v};
            for (int idx = 0; idx < 3; ++idx) {
                if (shp(sp->out[idx]) != Shape{1, s5[0], heads, Ntok, hd}) fail(sp, "unexpected qkv split sizes");
                const Node* sq = only_user(sp->out[idx]);
                if (!sq || sq->op != "Reshape" || shp(sq->out[0]) != Shape{s5[0], heads, Ntok, hd}) fail(sp, "expected the split outputs to be squeezed");
                *dst[idx] = sq->out[0];
                done.insert(sq);
            }
            done.insert(sp);
        } else {
            if (gs.size() != 3) fail(t5, "expected three q/k/v gathers");
            for (auto* gn : gs) {
                if (gn->op != "Gather" || gn->ai("axis", 0) != 0 || !is_c(gn->in[1])) fail(gn, "unexpected qkv split");
                int64_t idx = g.cst(gn->in[1]).i[0];
                (idx == 0 ? q : idx == 1 ? k : v) = gn->out[0];
                done.insert(gn);
            }
        }
        if (q.empty() || k.empty() || v.empty()) fail(t5, "q/k/v split incomplete");
        done.insert(r5); done.insert(t5);
        float scale = 1.f;
        std::string dyn; const HTensor* c = nullptr;
        // a scalar factor on q, on k (in front of or behind its transpose) or on the product: a decomposed scaled_dot_product_attention multiplies q AND k^T
        // by sqrt(scale); all of them end up in one factor on the scores
        auto absorb_scale = [&](std::string& name) {
            const Node* m = only_user(name);
            // (a Div only with the constant as the divisor: c / name is not a scale of name and leaves the pattern unmatched -> refusal)
            if (m && (m->op == "Mul" || (m->op == "Div" && m->in[0] == name)) && split_binary(m, dyn, c) && dyn == name && c->numel() == 1 && c->is_float()) {
                scale *= m->op == "Mul" ? c->f[0] : 1.f / c->f[0]; done.insert(m); name = m->out[0];
            }
        };
        absorb_scale(q);
        absorb_scale(k);
        const Node* kt = only_user(k);
        if (!kt || kt->op != "Transpose" || perm_of(kt) != std::vector<int64_t>{0, 1, 3, 2}) fail(kt ? kt : t5, "expected k transpose");
        done.insert(kt);
        std::string ktn = kt->out[0];
        absorb_scale(ktn);
        const Node* mm1 = only_user(q);
        if (!mm1 || mm1->op != "MatMul" || mm1->in[0] != q || mm1->in[1] != ktn) fail(mm1 ? mm1 : kt, "expected q @ k^T");
        done.insert(mm1);
        std::string cur = mm1->out[0];
        absorb_scale(cur);                        // optional scale after the product
        const Node* u = nullptr;
        std::vector<float> bias((size_t)heads * Ntok * Ntok, 0.f);
        u = only_user(cur);
        if (u && u->op == "Add" && split_binary(u, dyn, c)) {
            if (c->shape != Shape{1, heads, Ntok, Ntok} && c->shape != Shape{heads, Ntok, Ntok}) fail(u, "unexpected attention bias shape " + shape_str(c->shape));
            for (size_t e = 0; e < bias.size(); ++e) bias[e] = c->f[e];
            done.insert(u); cur = u->out[0];
        }
        // optional shifted-window mask: Reshape[B,nW,h,N,N] -> Add(const[1,nW,1,N,N]) -> Reshape
        std::vector<int32_t> maskid(nwin, 0);
        std::vector<std::vector<float>> masks(1, std::vector<float>((size_t)Ntok * Ntok, 0.f));
        u = only_user(cur);
        if (u && u->op == "Reshape" && shp(u->out[0]) == Shape{plan.B, nwin, heads, Ntok, Ntok}) {
            const Node* a = only_user(u->out[0]);
            if (!a || a->op != "Add" || !split_binary(a, dyn, c)) fail(u, "expected mask add");
            if (c->shape != Shape{1, nwin, 1, Ntok, Ntok} && c->shape != Shape{nwin, 1, Ntok, Ntok}) fail(a, "unexpected mask shape " + shape_str(c->shape));
            const Node* r = only_user(a->out[0]);
            if (!r || r->op != "Reshape" || shp(r->out[0]) != Shape{(int64_t)plan.B * nwin, heads, Ntok, Ntok}) fail(a, "expected reshape after mask");
            for (int wdx = 0; wdx < nwin; ++wdx) {
                std::vector<float> m(c->f.begin() + (size_t)wdx * Ntok * Ntok, c->f.begin() + (size_t)(wdx + 1) * Ntok * Ntok);
                int id = -1;
                for (size_t e = 0; e < masks.size(); ++e) if (masks[e] == m) { id = (int)e; break; }
                if (id < 0) { masks.push_back(m); id = (int)masks.size() - 1; }
                maskid[wdx] = id;
            }
            for (auto* z : {u, a, r}) done.insert(z);
            cur = r->out[0];
        }
        const Node* sm = only_user(cur);
        if (!sm || sm->op != "Softmax") fail(sm ? sm : mm1, "expected softmax");
        { int64_t ax = sm->ai("axis", -1); if (ax != -1 && ax != 3) fail(sm, "softmax axis"); }
        done.insert(sm);
        const Node* mm2 = only_user(sm->out[0]);
        if (!mm2 || mm2->op != "MatMul" || mm2->in[1] != v) fail(mm2 ? mm2 : sm, "expected attn @ v");
        done.insert(mm2);
        const Node* tr = only_user(mm2->out[0]);
        if (!tr || tr->op != "Transpose" || perm_of(tr) != std::vector<int64_t>{0, 2, 1, 3}) fail(mm2, "expected head merge transpose");
        const Node* rs = only_user(tr->out[0]);
        if (!rs || rs->op != "Reshape" || shp(rs->out[0]) != Shape{(int64_t)plan.B * nwin, Ntok, C}) fail(tr, "expected head merge reshape");
        done.insert(tr); done.insert(rs);

        if (plan.elt == 2 && !attn_supported(hd, Ntok))
            fail(mm1, "attention with " + std::to_string(Ntok) + " tokens per window and heads of " + std::to_string(hd) + " is not covered by the attention kernels (windows of 6x6 / 8x8, head sizes 8..64 in steps of 8)");
        Op op; op.kind = OP_ATTN; op.name = mm1->name.empty() ? "attention" : mm1->name;
        AttnOp& a = op.at;
        a.qkv = x.v.t; a.heads = heads; a.hd = hd; a.ws = ws; a.nwin = nwin; a.scale = scale;
        a.out = new_tensor(plan.B, x.H, x.W, C);
        a.nmask = (int)masks.size();
        std::vector<float> bm((size_t)a.nmask * heads * Ntok * Ntok);
        for (int m = 0; m < a.nmask; ++m) for (int h = 0; h < heads; ++h) for (int e = 0; e < Ntok * Ntok; ++e)
            bm[((size_t)m * heads + h) * Ntok * Ntok + e] = bias[(size_t)h * Ntok * Ntok + e] + masks[m][e];
        a.bias = blob_w(bm);
        a.maskid = blob_i32(maskid);
        op.flops = 2.0 * 2.0 * (double)plan.B * nwin * heads * Ntok * Ntok * hd;
        plan.ops.push_back(op);
        tensor_producer[a.out] = (int)plan.ops.size() - 1;
        LVal o; o.kind = LVal::WINROWS; o.v.t = a.out; o.v.H = x.H; o.v.W = x.W; o.C = C; o.ws = ws; o.H = x.H; o.W = x.W; o.table = x.table;
        vals[rs->out[0]] = o;
    }

    // ---- cunet squeeze-excite: mean(H,W) -> 1x1 conv -> relu -> 1x1 conv -> sigmoid -> x * s
    std::map<int, std::string> gate_names;
    // in-place scaling pass for a map whose gate is still pending
    void apply_gate(int t) {
        auto it = pending_gate.find(t);
        if (it == pending_gate.end()) return;
        Op sc; sc.kind = OP_SCALE_ADD; sc.name = gate_names[t];
        sc.se.scale = it->second; sc.se.pool = t; sc.se.C = plan.tensors[t].C;
        plan.ops.push_back(sc);
        pending_gate.erase(it);
    }

    bool try_lower_se(const Node* n) {
        if (!(n->op == "GlobalAveragePool" || n->op == "ReduceMean")) return false;
        const LVal& x = vals.at(n->in[0]);
        if (x.kind != LVal::MAP || !x.nchw) fail(n, "squeeze-excite input must be an NCHW map");
        if (n->op == "ReduceMean") {
            std::vector<int64_t> axes = n->has("axes") ? n->aints("axes") : (n->in.size() > 1 ? g.cst(n->in[1]).i : std::vector<int64_t>{});
            for (auto& a : axes) if (a < 0) a += 4;
            std::sort(axes.begin(), axes.end());
            if (axes != std::vector<int64_t>{2, 3} || n->ai("keepdims", 1) != 1) fail(n, "only spatial mean with keepdims is supported");
        }
        const Node* c1 = only_user(n->out[0]);
        if (!c1 || c1->op != "Conv") fail(n, "expected 1x1 conv after pooling");
        const Node* r = only_user(c1->out[0]);
        if (!r || r->op != "Relu") fail(c1, "expected relu");
        const Node* c2 = only_user(r->out[0]);
        if (!c2 || c2->op != "Conv") fail(r, "expected second 1x1 conv");
        const Node* sg = only_user(c2->out[0]);
        if (!sg || sg->op != "Sigmoid") fail(c2, "expected sigmoid");
        const Node* mul = only_user(sg->out[0]);
        if (!mul || mul->op != "Mul" || !((mul->in[0] == n->in[0] && mul->in[1] == sg->out[0]) || (mul->in[1] == n->in[0] && mul->in[0] == sg->out[0])))
            fail(sg, "expected channel scaling multiply");
        auto us = users(n->in[0]);
        if (us.size() != 2) fail(n, "squeeze-excite input has other consumers");
        const HTensor &w1 = g.cst(c1->in[1]), &w2 = g.cst(c2->in[1]);
        int C = x.C, Cm = (int)w1.shape[0];
        if (w1.shape != Shape{Cm, C, 1, 1} || w2.shape != Shape{C, Cm, 1, 1}) fail(c1, "unexpected squeeze-excite weights");
        auto it = tensor_producer.find(x.v.t);
        if (it == tensor_producer.end() || plan.ops[it->second].kind != OP_GEMM) fail(n, "squeeze-excite input is not produced by a fused op");
        const TensorDesc td = plan.tensors[x.v.t];      // by value: new_tensor() below grows plan.tensors (found by `make asan`)
        if (x.v.y0 || x.v.x0 || x.v.H != td.H || x.v.W != td.W) fail(n, "squeeze-excite over a cropped view");
        Op& prod = plan.ops[it->second];
        int nblocks = (prod.g.Mrows + 127) / 128;   // kGemmBM rows per workgroup, tiles never straddle batch items
        int pool = new_tensor(plan.B, 1, nblocks, td.C, 4);
        prod.g.pool_out = pool;
        Op op; op.kind = OP_SE; op.name = n->name.empty() ? "se" : n->name;
        SeOp& s = op.se;
        s.pool = pool; s.scale = new_tensor(plan.B, 1, 1, td.C, 4); s.C = C; s.Cmid = Cm;
        s.inv_count = 1.f / (float)(x.v.H * x.v.W);
        s.nblocks = nblocks; s.Mrows = prod.g.Mrows;
        s.w1 = blob_f32(w1.f); s.w2 = blob_f32(w2.f);
        std::vector<float> b1(Cm, 0.f), b2(C, 0.f);
        if (c1->in.size() > 2 && !c1->in[2].empty()) b1 = g.cst(c1->in[2]).f;
        if (c2->in.size() > 2 && !c2->in[2].empty()) b2 = g.cst(c2->in[2]).f;
        s.b1 = blob_f32(b1); s.b2 = blob_f32(b2);
        op.flops = 2.0 * plan.B * (2.0 * C * Cm);
        plan.ops.push_back(op);
        // The gate multiplies the whole map.  Its consumers in these graphs are 1x1 / 2x2 (transposed) convolutions and skip adds,
        // which take the multiplier on their operand load (same fp16 rounding as a separate pass); anything else gets the
        // in-place pass first (apply_gate).  Saves a read + write of the map per squeeze-excite block.
        gate_names[x.v.t] = mul->name.empty() ? "se_scale" : mul->name;
        pending_gate[x.v.t] = s.scale;
        if (plan.elt != 2 || switches().no_se_fold) apply_gate(x.v.t);
        for (auto* z : {n, c1, r, c2, sg, mul}) done.insert(z);
        vals[mul->out[0]] = x;
        return true;
    }

    // ---- post-pass: LN+qkv GEMM (window gather), attention core, proj GEMM (window scatter + residual) -> one op (k_swinattn.hip)
    void fuse_attn() {
        std::vector<Op> out;
        for (size_t i = 0; i < plan.ops.size(); ++i) {
            bool fused = false;
            if (i + 2 < plan.ops.size() && plan.ops[i].kind == OP_GEMM && plan.ops[i + 1].kind == OP_ATTN && plan.ops[i + 2].kind == OP_GEMM) {
                const GemmOp& q = plan.ops[i].g; const AttnOp& a = plan.ops[i + 1].at; const GemmOp& pr = plan.ops[i + 2].g;
                const int C = q.K;
                auto plain = [&](const View& v) { const TensorDesc& t = plan.tensors[v.t]; return v.y0 == 0 && v.x0 == 0 && v.H == t.H && v.W == t.W; };
                if (swin_attn_supported(C, a.heads, a.hd, a.ws) && q.amode == A_WIN && q.ln && q.act == ACT_NONE && q.omode == O_ROWS && q.N == 3 * C &&
                    q.res.t < 0 && !q.has_clip && q.stats_out < 0 && plain(q.a) && plan.tensors[q.a.t].C == C && a.qkv == q.out.t &&
                    pr.amode == A_ROWS && pr.a.t == a.out && !pr.ln && pr.act == ACT_NONE && pr.omode == O_WIN && pr.win_table == q.win_table &&
                    pr.K == C && pr.N == C && pr.res.t == q.a.t && pr.res2.t < 0 && !pr.has_clip && plain(pr.res) && pr.out.t != plan.out_tensor) {
                    Op m; m.kind = OP_SWINATTN; m.name = plan.ops[i].name + "+attn+" + plan.ops[i + 2].name;
                    m.flops = plan.ops[i].flops + plan.ops[i + 1].flops + plan.ops[i + 2].flops;
                    SwinAttnOp& s = m.sa;
                    s.x = q.a.t; s.y = pr.out.t; s.C = C; s.heads = a.heads; s.hd = a.hd; s.ws = a.ws; s.nwin = a.nwin; s.table = q.win_table;
                    s.wqkv = q.w; s.bqkv = q.bias; s.wproj = pr.w; s.bproj = pr.bias; s.maskid = a.maskid; s.scale = a.scale;
                    {   // fp32 bias table for the fused kernel, * log2(e), stored in the kernel's load order so every wave load is one
                        // contiguous 1 KB (or 256 B) block: [nmask][heads][query tile 3][ A: key tile 0 [64 lanes][4] | B: key tile 1
                        // [64 lanes][4] | C: keys 32..35 [64 lanes][1] ],  lane = 16*g + fr <-> query 16*qt + fr, keys 16*kt + 4*g + j.
                        const auto& src = plan.blobs[a.bias].data;
                        const int ntok = a.ws * a.ws;
                        const int per_qt = 64 * 4 * 2 + 64, per_unit = 3 * per_qt;
                        std::vector<float> t((size_t)a.nmask * a.heads * per_unit, 0.f);
                        auto at = [&](int m2, int qq, int kk) -> float {
                            if (qq >= ntok) qq = ntok - 1;
                            uint16_t hbits; memcpy(&hbits, &src[(((size_t)m2 * ntok + qq) * ntok + kk) * 2], 2);
                            return f16_to_f32(hbits) * 1.44269504088896341f;
                        };
                        for (int m2 = 0; m2 < a.nmask * a.heads; ++m2) for (int qt = 0; qt < 3; ++qt) for (int lane = 0; lane < 64; ++lane) {
                            const int fr = lane & 15, gg = lane >> 4, qq = qt * 16 + fr;
                            float* base = &t[(size_t)m2 * per_unit + (size_t)qt * per_qt];
                            for (int kt = 0; kt < 2; ++kt) for (int j = 0; j < 4; ++j) base[kt * 256 + lane * 4 + j] = at(m2, qq, kt * 16 + gg * 4 + j);
                            base[512 + lane] = at(m2, qq, 32 + gg);
                        }
                        s.bias = blob_f32(t);
                    }
                    s.eps = plan.ops[tensor_producer.at(q.a.t)].kind == OP_GEMM ? plan.ops[tensor_producer.at(q.a.t)].g.ln_eps : 1e-5f;
                    s.stats_out = pr.stats_out; s.eps_out = pr.ln_eps;
                    {   // closed form of the window table (shift + partition), verified entry by entry
                        const TensorDesc& xd = plan.tensors[q.a.t];
                        const int32_t* tb = (const int32_t*)plan.blobs[q.win_table].data.data();
                        s.H = xd.H; s.W = xd.W; s.ry = tb[0] / xd.W; s.rx = tb[0] % xd.W;
                        const int nwx = xd.W / a.ws;
                        for (int y = 0; y < xd.H && s.ry >= 0; ++y) for (int x = 0; x < xd.W; ++x) {
                            const int row = (((y / a.ws) * nwx + x / a.ws) * a.ws + y % a.ws) * a.ws + x % a.ws;
                            if (tb[row] != ((y + s.ry) % xd.H) * xd.W + (x + s.rx) % xd.W) { s.ry = s.rx = -1; break; }
                        }
                    }
                    out.push_back(m);
                    i += 2; fused = true;
                }
            }
            if (!fused) out.push_back(plan.ops[i]);
        }
        // tensor_producer indexes the old op list: rebuild it for the passes that follow
        plan.ops.swap(out);
        tensor_producer.clear();
        for (size_t i = 0; i < plan.ops.size(); ++i) {
            const Op& op = plan.ops[i];
            if (op.kind == OP_GEMM) tensor_producer[op.g.out.t] = (int)i;
            else if (op.kind == OP_ATTN) tensor_producer[op.at.out] = (int)i;
            else if (op.kind == OP_SWINATTN) tensor_producer[op.sa.y] = (int)i;
        }
    }

    // LayerNorm eps recorded by whichever op emits the statistics of tensor t
    float eps_of(int t) {
        auto it = tensor_producer.find(t);
        if (it == tensor_producer.end()) return 1e-5f;
        const Op& op = plan.ops[it->second];
        return op.kind == OP_GEMM ? op.g.ln_eps : op.kind == OP_SWINATTN ? op.sa.eps_out : op.kind == OP_MLP ? op.m.eps_out : 1e-5f;
    }

    // ---- post-pass: LN+fc1+GELU GEMM followed by fc2+residual GEMM  ->  one fused MLP op (k_mlp.hip)
    void fuse_mlp() {
        std::vector<Op> out;
        for (size_t i = 0; i < plan.ops.size(); ++i) {
            const Op& a = plan.ops[i];
            bool fused = false;
            if (i + 1 < plan.ops.size() && a.kind == OP_GEMM && plan.ops[i + 1].kind == OP_GEMM) {
                const GemmOp& g1 = a.g; const GemmOp& g2 = plan.ops[i + 1].g;
                const int C = g1.K;
                auto plain = [&](const View& v) { const TensorDesc& t = plan.tensors[v.t]; return v.y0 == 0 && v.x0 == 0 && v.H == t.H && v.W == t.W; };
                int hidden_users = 0;
                for (auto& o : plan.ops) if (o.kind == OP_GEMM && (o.g.a.t == g1.out.t || o.g.res.t == g1.out.t || o.g.res2.t == g1.out.t)) ++hidden_users;
                if (mlp_supported(C) && g1.amode == A_ROWS && g1.ln && g1.act == ACT_GELU && g1.omode == O_ROWS && g1.res.t < 0 && !g1.has_clip &&
                    g1.N == 2 * C && g1.stats_out < 0 && g1.pool_out < 0 && plain(g1.a) && plan.tensors[g1.a.t].C == C &&
                    g2.amode == A_ROWS && !g2.ln && g2.act == ACT_NONE && g2.omode == O_ROWS && g2.a.t == g1.out.t && g2.K == 2 * C && g2.N == C &&
                    g2.res.t == g1.a.t && g2.res2.t < 0 && !g2.has_clip && g2.pool_out < 0 && plain(g2.res) && hidden_users == 1 && g1.out.t != plan.out_tensor) {
                    Op m; m.kind = OP_MLP; m.name = a.name + "+" + plan.ops[i + 1].name; m.flops = a.flops + plan.ops[i + 1].flops;
                    m.m.x = g1.a.t; m.m.y = g2.out.t; m.m.C = C; m.m.w1 = g1.w; m.m.b1 = g1.bias; m.m.w2 = g2.w; m.m.b2 = g2.bias;
                    m.m.eps = eps_of(g1.a.t);
                    m.m.stats_out = g2.stats_out; m.m.eps_out = g2.ln_eps;
                    out.push_back(m);
                    ++i; fused = true;
                }
            }
            if (!fused) out.push_back(a);
        }
        plan.ops.swap(out);
        // the row statistics of a tensor that only fused MLP ops normalise are no longer needed
        for (auto& op : plan.ops) {
            int* so = op.kind == OP_GEMM ? &op.g.stats_out : op.kind == OP_MLP ? &op.m.stats_out : op.kind == OP_SWINATTN ? &op.sa.stats_out : nullptr;
            if (!so || *so < 0) continue;
            bool used = false;
            for (auto& o : plan.ops) if (o.kind == OP_GEMM && o.g.stats_in == *so) used = true;
            if (!used) *so = -1;
        }
    }

    // ---- LayerNorm as exporters write it below opset 17 (nvonnxparser takes either form, img2img_build.cpp:81-88):
    //          m = ReduceMean(x, axes=[-1]);  d = Sub(x, m);  v = ReduceMean(Pow(d, 2) | Mul(d, d), axes=[-1]);
    //          y = Div(d, Sqrt(Add(v, eps)))  [ * gamma ]  [ + beta ]
    //      -> the same LVal the LayerNormalization node gives (normalisation folded into the consuming product)
    bool try_lower_decomposed_layernorm(const Node* n) {
        auto itx = vals.find(n->in[0]);
        if (itx == vals.end()) return false;
        const LVal x = itx->second;
        if (x.kind != LVal::MAP || x.nchw || x.ln) return false;
        auto last_axis = [&](const Node* r) {
            std::vector<int64_t> ax = r->aints("axes");
            if (ax.empty() && r->in.size() > 1 && is_c(r->in[1])) ax = g.cst(r->in[1]).i;          // opset 18: axes as an input
            return ax.size() == 1 && (ax[0] == -1 || ax[0] == 3) && r->ai("keepdims", 1) == 1;
        };
        if (!last_axis(n)) return false;
        const Node* sub = nullptr;
        for (auto* u : users(n->out[0])) if (u->op == "Sub" && u->in.size() == 2 && u->in[0] == n->in[0] && u->in[1] == n->out[0]) sub = u;
        if (!sub || users(n->out[0]).size() != 1) return false;          // the mean feeds the subtraction only (x itself may go on to a residual add)
        const Node *sq = nullptr, *div = nullptr;
        for (auto* u : users(sub->out[0])) {
            const HTensor* c = nullptr; std::string dyn;
            if (u->op == "Pow" && u->in[0] == sub->out[0] && is_c(u->in[1]) && scalar_near(&g.cst(u->in[1]), 2.0)) sq = u;
            else if (u->op == "Mul" && u->in[0] == sub->out[0] && u->in[1] == sub->out[0]) sq = u;
            else if (u->op == "Div" && u->in[0] == sub->out[0]) div = u;
            (void)c; (void)dyn;
        }
        if (!sq || !div || users(sub->out[0]).size() != (sq->op == "Mul" ? 3u : 2u)) return false;
        const Node* var = only_user(sq->out[0]);
        if (!var || var->op != "ReduceMean" || !last_axis(var)) return false;
        const Node* add = only_user(var->out[0]);
        std::string dyn; const HTensor* ceps = nullptr;
        if (!add || add->op != "Add" || !split_binary(add, dyn, ceps) || dyn != var->out[0] || ceps->numel() != 1 || !ceps->is_float()) return false;
        const Node* sqrt = only_user(add->out[0]);
        if (!sqrt || sqrt->op != "Sqrt" || only_user(sqrt->out[0]) != div || div->in[1] != sqrt->out[0]) return false;
        LVal y = x; y.ln = true; y.gamma = nullptr; y.beta = nullptr; y.eps = ceps->f[0];
        std::vector<const Node*> used = {n, sub, sq, var, add, sqrt, div};
        std::string out = div->out[0];
        // optional affine part: Mul by gamma [C], Add of beta [C]
        const Node* mul = only_user(out);
        const HTensor* cg = nullptr;
        if (mul && mul->op == "Mul" && split_binary(mul, dyn, cg) && dyn == out && (int)cg->numel() == x.C && cg->is_float()) {
            y.gamma = cg; used.push_back(mul); out = mul->out[0];
            const Node* addb = only_user(out);
            const HTensor* cb = nullptr;
            if (addb && addb->op == "Add" && split_binary(addb, dyn, cb) && dyn == out && (int)cb->numel() == x.C && cb->is_float()) { y.beta = cb; used.push_back(addb); out = addb->out[0]; }
        }
        if (!y.gamma) { ones_c.emplace_back(); HTensor& o = ones_c.back(); o.shape = {x.C}; o.f.assign(x.C, 1.f); y.gamma = &o; }     // no affine part: gamma = 1
        for (auto* z : used) done.insert(z);
        vals[out] = y;
        return true;
    }
    std::list<HTensor> ones_c;   // constants synthesised by the lowering (stable addresses)

    // ---------------------------------------------------------------------------------------------------------
    Plan run() {
        const Shape& is = shp(g.input);
        plan.B = (int)is[0]; plan.Cin = (int)is[1]; plan.T = (int)is[2];
        if (is[2] != is[3]) throw std::runtime_error("only square tiles are supported");
        if (plan.Cin != 3) throw std::runtime_error("expected a 3-channel input");
        plan.in_tensor = new_tensor(plan.B, plan.T, plan.T, 4);
        { LVal v; v.v.t = plan.in_tensor; v.v.H = v.v.W = plan.T; v.C = 3; v.nchw = true; vals[g.input] = v; }

        for (const Node* n : g.nodes) {
            if (done.count(n)) continue;
            const std::string& op = n->op;
            if (op == "Conv" || op == "ConvTranspose") { lower_conv(n); continue; }
            if (op == "MatMul") { lower_matmul(n); continue; }
            if (op == "ReduceMean" && try_lower_decomposed_layernorm(n)) continue;
            if (op == "GlobalAveragePool" || op == "ReduceMean") { if (try_lower_se(n)) continue; }
            auto itx = vals.find(n->in[0]);
            if (itx == vals.end()) fail(n, "input \"" + n->in[0] + "\" was not lowered");
            const LVal x = itx->second;
            if (op == "Pad") { lower_pad(n, x); continue; }
            if (op != "Slice")   // crops are views; every other consumer of a gated map gets the in-place pass first
                for (const std::string& in : n->in) { auto iv = vals.find(in); if (iv != vals.end() && iv->second.kind == LVal::MAP) apply_gate(iv->second.v.t); }
            if (op == "Slice" && x.kind == LVal::MAP && n->in.size() > 2 && is_c(n->in[1]) && is_c(n->in[2])) {
                // a Slice that keeps everything is the identity in any layout (torchvision's shifted_window_attention ends with x[:, :H, :W, :])
                std::vector<int64_t> starts = g.cst(n->in[1]).i, ends = g.cst(n->in[2]).i;
                std::vector<int64_t> axes = n->in.size() > 3 && !n->in[3].empty() ? g.cst(n->in[3]).i : std::vector<int64_t>{};
                std::vector<int64_t> steps = n->in.size() > 4 && !n->in[4].empty() ? g.cst(n->in[4]).i : std::vector<int64_t>{};
                const int64_t dims[4] = {plan.B, x.nchw ? x.C : x.v.H, x.nchw ? x.v.H : x.v.W, x.nchw ? x.v.W : x.C};
                bool whole = !starts.empty() && starts.size() == ends.size();
                for (size_t k = 0; k < starts.size() && whole; ++k) {
                    int64_t ax = axes.empty() ? (int64_t)k : axes[k]; if (ax < 0) ax += 4;
                    if (ax < 0 || ax > 3 || (!steps.empty() && steps[k] != 1)) { whole = false; break; }
                    const int64_t D = dims[ax], st = starts[k] < 0 ? starts[k] + D : starts[k], en = ends[k] < 0 ? ends[k] + D : std::min<int64_t>(ends[k], D);
                    whole = st == 0 && en == D;
                }
                if (whole) { vals[n->out[0]] = x; done.insert(n); continue; }
            }
            if (op == "Slice" && x.kind == LVal::MAP && x.nchw && !x.ln && true) {
                // crop expressed as Slice on H/W
                std::vector<int64_t> starts = g.cst(n->in[1]).i, ends = g.cst(n->in[2]).i;
                std::vector<int64_t> axes = n->in.size() > 3 && !n->in[3].empty() ? g.cst(n->in[3]).i : std::vector<int64_t>{};
                LVal y = x; bool ok = true;
                for (size_t k = 0; k < starts.size(); ++k) {
                    int64_t ax = axes.empty() ? (int64_t)k : axes[k]; if (ax < 0) ax += 4;
                    int64_t D = ax == 2 ? x.v.H : ax == 3 ? x.v.W : -1; if (D < 0) { ok = false; break; }
                    int64_t s = starts[k] < 0 ? starts[k] + D : starts[k], e = ends[k] < 0 ? ends[k] + D : std::min<int64_t>(ends[k], D);
                    if (ax == 2) { y.v.y0 += (int)s; y.v.H = (int)(e - s); } else { y.v.x0 += (int)s; y.v.W = (int)(e - s); }
                }
                if (ok) { vals[n->out[0]] = y; done.insert(n); continue; }
            }
            if (op == "Transpose") {
                auto perm = perm_of(n);
                if (x.kind == LVal::MAP && x.nchw && perm == std::vector<int64_t>{0, 2, 3, 1}) { LVal y = x; y.nchw = false; vals[n->out[0]] = y; done.insert(n); continue; }
                if (x.kind == LVal::MAP && !x.nchw && !x.ln && perm == std::vector<int64_t>{0, 3, 1, 2}) { LVal y = x; y.nchw = true; vals[n->out[0]] = y; done.insert(n); continue; }
                fail(n, "unsupported transpose");
            }
            if (op == "LayerNormalization") {
                if (x.kind != LVal::MAP || x.nchw || x.ln) fail(n, "LayerNorm input must be a plain NHWC map");
                int64_t ax = n->ai("axis", -1);
                if (ax != -1 && ax != 3) fail(n, "LayerNorm must normalise the channel axis");
                LVal y = x; y.ln = true; y.gamma = &g.cst(n->in[1]); y.beta = n->in.size() > 2 && !n->in[2].empty() ? &g.cst(n->in[2]) : nullptr;
                y.eps = n->af("epsilon", 1e-5f);
                if ((int)y.gamma->numel() != x.C) fail(n, "LayerNorm width mismatch");
                vals[n->out[0]] = y; done.insert(n); continue;
            }
            if (op == "Slice" && x.kind == LVal::MAP && !x.nchw) {
                // cyclic shift: handled from the first slice of the pair
                int ax, sh; std::string out; std::vector<const Node*> used;
                if (!try_roll_axis(n->in[0], ax, sh, out, used)) fail(n, "unsupported slice pattern");
                LVal y = x;
                if (ax == 1) y.roll_y = (y.roll_y + sh) % x.v.H; else if (ax == 2) y.roll_x = (y.roll_x + sh) % x.v.W; else fail(n, "roll on unsupported axis");
                for (auto* z : used) done.insert(z);
                vals[out] = y; continue;
            }
            if (op == "Reshape") {
                const Shape& os = shp(n->out[0]);
                if (x.kind == LVal::MAP && !x.nchw && os.size() == 6) {
                    // window partition: [B,H/ws,ws,W/ws,ws,C] -> Transpose(0,1,3,2,4,5) -> [B*nW, ws*ws, C]
                    int ws = (int)os[2];
                    if (os != Shape{plan.B, x.v.H / ws, ws, x.v.W / ws, ws, x.C} || x.v.H % ws || x.v.W % ws) fail(n, "unexpected window partition " + shape_str(os));
                    const Node* t = only_user(n->out[0]);
                    if (!t || t->op != "Transpose" || perm_of(t) != std::vector<int64_t>{0, 1, 3, 2, 4, 5}) fail(n, "unexpected window partition permutation");
                    const Node* r2 = only_user(t->out[0]);
                    int nwin = (x.v.H / ws) * (x.v.W / ws);
                    if (!r2 || r2->op != "Reshape" || shp(r2->out[0]) != Shape{(int64_t)plan.B * nwin, ws * ws, x.C}) fail(t, "unexpected window partition reshape");
                    const TensorDesc& td = plan.tensors[x.v.t];
                    if (x.v.y0 || x.v.x0 || x.v.H != td.H || x.v.W != td.W) fail(n, "window partition of a cropped view");
                    LVal y = x; y.ws = ws; y.table = window_table(x.v.H, x.v.W, ws, x.roll_y, x.roll_x, false);
                    y.roll_x = y.roll_y = 0;
                    for (auto* z : {n, t, r2}) done.insert(z);
                    vals[r2->out[0]] = y; continue;
                }
                if (x.kind == LVal::WINROWS && os.size() == 5) { lower_attention(n); continue; }
                fail(n, "unsupported reshape " + shape_str(os));
            }
            fail(n, "no lowering rule (input kind " + std::to_string((int)x.kind) + ")");
        }
        auto it = vals.find(g.output);
        if (it == vals.end()) throw std::runtime_error("graph output was not lowered");
        const LVal& y = it->second;
        const Shape& os = shp(g.output);
        if (y.kind != LVal::MAP || !y.nchw || y.ln || y.C != 3 || os[2] != os[3]) throw std::runtime_error("graph output must be a plain [B,3,T',T'] map");
        const TensorDesc& td = plan.tensors[y.v.t];
        if (y.v.y0 || y.v.x0 || y.v.H != td.H || y.v.W != td.W) throw std::runtime_error("graph output is a cropped view");
        plan.out_tensor = y.v.t; plan.Tout = (int)os[2]; plan.Cout = 3;
        if (!switches().no_fuse && plan.elt == 2) { if (!switches().no_fuse_attn) fuse_attn(); fuse_mlp(); }   // the fused kernels are fp16 kernels
        for (auto& op : plan.ops) plan.flops += op.flops;
        bool has_attn = false; for (auto& op : plan.ops) has_attn |= op.kind == OP_ATTN || op.kind == OP_SWINATTN;
        plan.model_kind = has_attn ? "swin_unet" : "cunet";
        return plan;
    }
};

}  // namespace

Plan lower_graph(const FoldedGraph& g, bool fp32) {
    Lowerer l(g);
    l.plan.elt = fp32 ? 4 : 2;
    return l.run();
}

// "opset 17 (pytorch), 6712 nodes: Conv x2, MatMul x56, ..." - what the parser was handed (img2img_build.cpp:81-88 hands any ONNX file
// to TensorRT's parser, which lists the layers it made of it); build() logs it so that a graph this loader was not written for shows
// what it consists of before any node is refused
std::string onnx_op_histogram(const std::string& onnx_path) {
    const Model m = load_onnx(onnx_path);
    std::map<std::string, int> count;
    for (const Node& n : m.nodes) ++count[n.op];
    std::vector<std::pair<int, std::string>> by;
    for (const auto& kv : count) by.push_back({-kv.second, kv.first});
    std::sort(by.begin(), by.end());
    std::string s = "opset " + std::to_string(m.opset) + (m.producer.empty() ? "" : " (" + m.producer + ")") + ", " + std::to_string(m.nodes.size()) + " nodes:";
    for (const auto& e : by) s += " " + e.second + " x" + std::to_string(-e.first);
    return s;
}

Plan build_plan(const std::string& onnx_path, int batch, int channels, int height, int width, bool fp32) {
    Model m = load_onnx(onnx_path);
    FoldedGraph g = fold_graph(m, {batch, channels, height, width});
    return lower_graph(g, fp32);
}

}  // namespace w2x
