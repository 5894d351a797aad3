#include "plan.h"

#include <algorithm>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <type_traits>

namespace w2x {

static_assert(std::is_trivially_copyable<GemmOp>::value, "GemmOp must be POD for serialization");
static_assert(std::is_trivially_copyable<AttnOp>::value, "AttnOp must be POD for serialization");
static_assert(std::is_trivially_copyable<SeOp>::value, "SeOp must be POD for serialization");

std::string Plan::describe() const {
    std::ostringstream o;
    o << "plan " << model_kind << " in=[" << B << "," << Cin << "," << T << "," << T << "] out=[" << B << "," << Cout << "," << Tout << "," << Tout
      << "] ops=" << ops.size() << " tensors=" << tensors.size() << " blobs=" << blobs.size() << " flops=" << (long long)flops << " precision=" << (elt == 4 ? "fp32" : "fp16") << "\n";
    int64_t act = 0; for (auto& t : tensors) act += t.bytes();
    int64_t wb = 0; for (auto& b : blobs) wb += (int64_t)b.data.size();
    o << "activation_bytes=" << act << " constant_bytes=" << wb << "\n";
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op& op = ops[i];
        o << i << " ";
        if (op.kind == OP_GEMM) {
            const GemmOp& g = op.g;
            const char* am = g.amode == A_ROWS ? "rows" : g.amode == A_WIN ? "win" : "conv";
            const char* om = g.omode == O_ROWS ? "rows" : g.omode == O_WIN ? "win" : "pixshuf";
            o << "gemm a=" << am;
            if (g.amode == A_CONV) o << g.kh << "x" << g.kw << "s" << g.stride;
            o << " M=" << (int64_t)B * g.Mrows << " K=" << g.K << " N=" << g.N << " ln=" << g.ln << " act=" << g.act << " o=" << om;
            if (g.omode == O_PIXSHUF) o << g.r;
            if (g.se_scale >= 0) o << " a*gate";
            if (g.res_scale >= 0) o << " res*gate";
            o << " res=" << (g.res.t >= 0) + (g.res2.t >= 0) << " clip=" << g.has_clip << " stats=" << (g.stats_out >= 0) << " pool=" << (g.pool_out >= 0)
              << " t" << g.a.t << "->t" << g.out.t;
        } else if (op.kind == OP_ATTN) {
            const AttnOp& a = op.at;
            o << "attn heads=" << a.heads << " hd=" << a.hd << " ws=" << a.ws << " nwin=" << a.nwin << " nmask=" << a.nmask << " scale=" << a.scale << " t" << a.qkv << "->t" << a.out;
        } else if (op.kind == OP_SE) {
            o << "se C=" << op.se.C << " Cmid=" << op.se.Cmid;
        } else if (op.kind == OP_SCALE_ADD) {
            o << "scale t" << op.se.pool;
        } else if (op.kind == OP_SWINATTN) {
            o << "swinattn C=" << op.sa.C << " heads=" << op.sa.heads << " hd=" << op.sa.hd << " nwin=" << op.sa.nwin << " stats=" << (op.sa.stats_out >= 0) << " t" << op.sa.x << "->t" << op.sa.y;
        } else if (op.kind == OP_MLP) {
            o << "mlp C=" << op.m.C << " M=" << (int64_t)tensors[op.m.x].B * tensors[op.m.x].H * tensors[op.m.x].W << " stats=" << (op.m.stats_out >= 0) << " t" << op.m.x << "->t" << op.m.y;
        }
        {   // every tensor the op touches (the set the engine's lifetime analysis uses), for tools and tests
            std::vector<int> refs;
            switch (op.kind) {
                case OP_GEMM: refs = {op.g.a.t, op.g.res.t, op.g.res2.t, op.g.stats_in, op.g.se_scale, op.g.res_scale, op.g.out.t, op.g.stats_out, op.g.pool_out}; break;
                case OP_ATTN: refs = {op.at.qkv, op.at.out}; break;
                case OP_SE: case OP_SCALE_ADD: refs = {op.se.pool, op.se.scale}; break;
                case OP_MLP: refs = {op.m.x, op.m.y, op.m.stats_out}; break;
                case OP_SWINATTN: refs = {op.sa.x, op.sa.y, op.sa.stats_out}; break;
                default: break;
            }
            o << " refs=";
            bool first = true;
            for (int t : refs) if (t >= 0) { o << (first ? "" : ",") << "t" << t; first = false; }
        }
        o << " flops=" << (long long)op.flops << " [" << op.name << "]\n";
    }
    return o.str();
}

namespace {
struct Writer {
    std::vector<uint8_t> b;
    template <class T> void pod(const T& v) { size_t n = b.size(); b.resize(n + sizeof(T)); memcpy(&b[n], &v, sizeof(T)); }
    void str(const std::string& s) { pod<uint64_t>(s.size()); b.insert(b.end(), s.begin(), s.end()); }
    void bytes(const std::vector<uint8_t>& v) { pod<uint64_t>(v.size()); b.insert(b.end(), v.begin(), v.end()); }
};
struct Reader {
    const uint8_t* p; const uint8_t* e;
    template <class T> T pod() { if ((size_t)(e - p) < sizeof(T)) throw std::runtime_error("engine file truncated"); T v; memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    std::string str() { uint64_t n = pod<uint64_t>(); if ((uint64_t)(e - p) < n) throw std::runtime_error("engine file truncated"); std::string s((const char*)p, n); p += n; return s; }
    std::vector<uint8_t> bytes() { uint64_t n = pod<uint64_t>(); if ((uint64_t)(e - p) < n) throw std::runtime_error("engine file truncated"); std::vector<uint8_t> v(p, p + n); p += n; return v; }
};
constexpr uint64_t kMagic = 0x3158325755464957ull;  // "WIFUW2X1"
constexpr uint32_t kVersion = 10;
}  // namespace

std::vector<uint8_t> Plan::serialize() const {
    Writer w;
    w.pod(kMagic); w.pod(kVersion);
    w.pod<uint32_t>(sizeof(GemmOp)); w.pod<uint32_t>(sizeof(AttnOp)); w.pod<uint32_t>(sizeof(SeOp) + sizeof(MlpOp) + sizeof(SwinAttnOp));
    w.pod(B); w.pod(userB); w.pod(Cin); w.pod(T); w.pod(Tout); w.pod(Cout); w.pod(elt); w.pod(in_tensor); w.pod(out_tensor); w.pod(flops);
    w.str(model_kind);
    w.pod<uint64_t>(tensors.size()); for (auto& t : tensors) w.pod(t);
    w.pod<uint64_t>(blobs.size()); for (auto& b : blobs) w.bytes(b.data);
    w.pod<uint64_t>(ops.size());
    for (auto& op : ops) { w.pod(op.kind); w.str(op.name); w.pod(op.g); w.pod(op.at); w.pod(op.se); w.pod(op.m); w.pod(op.sa); w.pod(op.flops); }
    return std::move(w.b);
}

Plan Plan::deserialize(const uint8_t* p, size_t n) {
    Reader r{p, p + n};
    if (r.pod<uint64_t>() != kMagic) throw std::runtime_error("not a w2x engine file");
    if (r.pod<uint32_t>() != kVersion) throw std::runtime_error("engine file version mismatch");
    if (r.pod<uint32_t>() != sizeof(GemmOp) || r.pod<uint32_t>() != sizeof(AttnOp) || r.pod<uint32_t>() != sizeof(SeOp) + sizeof(MlpOp) + sizeof(SwinAttnOp)) throw std::runtime_error("engine file layout mismatch");
    Plan pl;
    pl.B = r.pod<int>(); pl.userB = r.pod<int>(); pl.Cin = r.pod<int>(); pl.T = r.pod<int>(); pl.Tout = r.pod<int>(); pl.Cout = r.pod<int>(); pl.elt = r.pod<int>();
    pl.in_tensor = r.pod<int>(); pl.out_tensor = r.pod<int>(); pl.flops = r.pod<double>();
    pl.model_kind = r.str();
    uint64_t nt = r.pod<uint64_t>(); pl.tensors.resize(nt); for (auto& t : pl.tensors) t = r.pod<TensorDesc>();
    uint64_t nb = r.pod<uint64_t>(); pl.blobs.resize(nb); for (auto& b : pl.blobs) b.data = r.bytes();
    uint64_t no = r.pod<uint64_t>(); pl.ops.resize(no);
    for (auto& op : pl.ops) { op.kind = r.pod<int>(); op.name = r.str(); op.g = r.pod<GemmOp>(); op.at = r.pod<AttnOp>(); op.se = r.pod<SeOp>(); op.m = r.pod<MlpOp>(); op.sa = r.pod<SwinAttnOp>(); op.flops = r.pod<double>(); }
    pl.validate();
    return pl;
}

// A plan file is an input from disk: every index the engine later uses to address tensors[], blobs[] or device memory is
// checked here, so that a stale, truncated or edited file ends as "Failed to deserialize engine" and never as a host
// out-of-bounds access or a device fault.
void Plan::validate() const {
    auto fail = [](const std::string& what) { throw std::runtime_error("engine file inconsistent: " + what); };
    const int nt = (int)tensors.size(), nb = (int)blobs.size();
    if (B <= 0 || userB <= 0 || B % userB || T <= 0 || Tout <= 0 || Cin != 3 || Cout != 3) fail("header");
    if (elt != 2 && elt != 4) fail("precision");
    if (in_tensor < 0 || in_tensor >= nt || out_tensor < 0 || out_tensor >= nt) fail("input / output tensor id");
    // every tensor carries the plan's B tile slots: the per-image strides of the kernels and the tile-group addressing of the arena
    // (a tensor of B tiles at offset o becomes B / NG tiles at o / NG, engine.cpp group_ptr) rest on it
    for (const TensorDesc& t : tensors) if (t.B != B || t.H <= 0 || t.W <= 0 || t.C <= 0 || (t.elt != 2 && t.elt != 4) || t.bytes() > ((int64_t)1 << 40)) fail("tensor shape");
    auto floats_per_tile = [&](int id) { const TensorDesc& d = tensors[id]; return (int64_t)d.H * d.W * d.C; };
    // a [pixels][2] fp32 side tensor with the LayerNorm statistics of the rows of map `of`
    auto stats_of = [&](int id, int of) {
        if (id < 0) return;
        const TensorDesc& d = tensors[id], &m = tensors[of];
        if (d.elt != 4 || (int64_t)d.H * d.W * d.C < (int64_t)m.H * m.W * 2) fail("statistics tensor shape");
    };
    if (tensors[in_tensor].H != T || tensors[in_tensor].W != T || tensors[in_tensor].C != 4 || tensors[in_tensor].B != B) fail("input tensor shape");
    if (tensors[out_tensor].H != Tout || tensors[out_tensor].W != Tout || tensors[out_tensor].C != 4) fail("output tensor shape");
    if (tensors[in_tensor].elt != elt || tensors[out_tensor].elt != elt) fail("input / output tensor precision");
    auto ten = [&](int id, bool optional) { if (id < (optional ? -1 : 0) || id >= nt) fail("tensor id"); };
    auto blob = [&](int id, bool optional, size_t bytes) {
        if (id < (optional ? -1 : 0) || id >= nb) fail("blob id");
        if (id >= 0 && bytes && blobs[id].data.size() < bytes) fail("blob size");
    };
    auto view = [&](const View& v, bool optional) {
        ten(v.t, optional);
        if (v.t < 0) return;
        const TensorDesc& d = tensors[v.t];
        if (v.H <= 0 || v.W <= 0 || v.y0 < 0 || v.x0 < 0 || v.y0 + v.H > d.H || v.x0 + v.W > d.W) fail("view outside its tensor");
    };
    for (const Op& op : ops) {
        switch (op.kind) {
            case OP_GEMM: {
                const GemmOp& g = op.g;
                view(g.res, true); view(g.res2, true); view(g.out, false);
                ten(g.a.t, false);
                ten(g.stats_in, true); ten(g.stats_out, true); ten(g.pool_out, true); ten(g.se_scale, true); ten(g.res_scale, true);
                if (g.res_scale >= 0 && g.res.t < 0) fail("residual gate without a residual");
                if (g.se_scale >= 0 && (tensors[g.se_scale].elt != 4 || (int64_t)tensors[g.se_scale].B * tensors[g.se_scale].H * tensors[g.se_scale].W * tensors[g.se_scale].C < (int64_t)B * tensors[g.a.t].C)) fail("input gate shape");
                if (g.res_scale >= 0 && (tensors[g.res_scale].elt != 4 || (int64_t)tensors[g.res_scale].B * tensors[g.res_scale].H * tensors[g.res_scale].W * tensors[g.res_scale].C < (int64_t)B * tensors[g.res.t].C)) fail("residual gate shape");
                if (g.K <= 0 || g.N <= 0 || g.Mrows <= 0 || g.kh <= 0 || g.kw <= 0 || g.stride <= 0 || g.r <= 0 || g.aW <= 0) fail("gemm shape");
                if (g.amode < A_ROWS || g.amode > A_CONV || g.omode < O_ROWS || g.omode > O_PIXSHUF) fail("gemm mode");
                if (g.K != g.kh * g.kw * tensors[g.a.t].C) fail("gemm K");
                if (g.Cout <= 0 || tensors[g.out.t].C < g.Cout || (g.res.t >= 0 && tensors[g.res.t].C < g.Cout) || (g.res2.t >= 0 && tensors[g.res2.t].C < g.Cout)) fail("gemm output channels");
                {   // the A side reads pixel (y0 + oy*stride + ky, x0 + ox*stride + kx) for the Mrows = oH x aW output positions
                    // (a crop folded into a convolution moves y0 / x0 without touching the view's H / W, so those are not used)
                    const TensorDesc& d = tensors[g.a.t];
                    const int oH = (g.Mrows + g.aW - 1) / g.aW;
                    if (g.a.y0 < 0 || g.a.x0 < 0 || (int64_t)g.a.y0 + (int64_t)(oH - 1) * g.stride + g.kh > d.H || (int64_t)g.a.x0 + (int64_t)(g.aW - 1) * g.stride + g.kw > d.W) fail("gemm input window outside its tensor");
                    if (g.amode == A_WIN && g.Mrows != d.H * d.W) fail("window gather geometry");
                }
                blob(g.w, false, (size_t)g.N * ((g.K + 7) / 8 * 8) * elt);
                if (tensors[g.a.t].elt != elt || tensors[g.out.t].elt != elt || (g.res.t >= 0 && tensors[g.res.t].elt != elt) || (g.res2.t >= 0 && tensors[g.res2.t].elt != elt)) fail("gemm operand precision");
                blob(g.bias, false, (size_t)g.N * 4);
                blob(g.csum, !g.ln, (size_t)g.N * 4);
                if (g.amode == A_WIN || g.omode == O_WIN) blob(g.win_table, false, (size_t)g.Mrows * 4); else blob(g.win_table, true, 0);
                if (g.ln && g.stats_in < 0) fail("LayerNorm without statistics");
                stats_of(g.stats_in, g.a.t); stats_of(g.stats_out, g.out.t);
                if (g.pool_out >= 0 && (tensors[g.pool_out].elt != 4 || floats_per_tile(g.pool_out) < (int64_t)((g.Mrows + 127) / 128) * tensors[g.out.t].C)) fail("pooled sums tensor shape");
                break;
            }
            case OP_ATTN: {
                const AttnOp& a = op.at;
                ten(a.qkv, false); ten(a.out, false);
                if (a.heads <= 0 || a.hd <= 0 || a.ws <= 0 || a.nwin <= 0 || a.nmask <= 0) fail("attention shape");
                const size_t n = (size_t)a.ws * a.ws;
                blob(a.bias, false, (size_t)a.nmask * a.heads * n * n * elt);
                if (tensors[a.qkv].elt != elt || tensors[a.out].elt != elt) fail("attention operand precision");
                {   // rows in window order: [nwin * ws^2][3 * heads * hd] -> [nwin * ws^2][heads * hd]
                    const TensorDesc& q = tensors[a.qkv], &o = tensors[a.out];
                    if (q.C != 3 * a.heads * a.hd || o.C != a.heads * a.hd || (int64_t)q.H * q.W != (int64_t)a.nwin * n || (int64_t)o.H * o.W != (int64_t)a.nwin * n) fail("attention tensor shape");
                }
                blob(a.maskid, false, (size_t)a.nwin * 4);
                for (int i = 0; i < a.nwin; ++i) { int m; memcpy(&m, blobs[a.maskid].data.data() + 4 * (size_t)i, 4); if (m < 0 || m >= a.nmask) fail("attention mask id"); }
                break;
            }
            case OP_SE: {
                const SeOp& e = op.se;
                ten(e.pool, false); ten(e.scale, false);
                if (e.C <= 0 || e.Cmid <= 0 || e.nblocks <= 0 || e.Mrows <= 0) fail("squeeze-excite shape");
                // pool: [B][nblocks][Cs] partial sums written by the producing convolution, scale: [B][Cs] gates
                if (tensors[e.pool].elt != 4 || tensors[e.scale].elt != 4 || tensors[e.pool].C < e.C || tensors[e.scale].C != tensors[e.pool].C ||
                    floats_per_tile(e.pool) < (int64_t)e.nblocks * tensors[e.pool].C || e.nblocks < (e.Mrows + 127) / 128) fail("squeeze-excite tensor shape");
                blob(e.w1, false, (size_t)e.C * e.Cmid * 4); blob(e.b1, false, (size_t)e.Cmid * 4);
                blob(e.w2, false, (size_t)e.C * e.Cmid * 4); blob(e.b2, false, (size_t)e.C * 4);
                break;
            }
            case OP_SCALE_ADD: {   // in-place gate: map (se.pool) *= gate (se.scale, fp32 [B][C]); rows move as 16-byte pieces
                ten(op.se.pool, false); ten(op.se.scale, false);
                const TensorDesc& m = tensors[op.se.pool], &gt = tensors[op.se.scale];
                if (m.elt != elt || gt.elt != 4 || m.C % (elt == 2 ? 8 : 4) || floats_per_tile(op.se.scale) < m.C) fail("gate pass shape");
                break;
            }
            case OP_MLP: {
                const MlpOp& m = op.m;
                ten(m.x, false); ten(m.y, false); ten(m.stats_out, true);
                if (m.C <= 0 || tensors[m.x].C != m.C || tensors[m.y].C != m.C) fail("MLP width");
                if (tensors[m.x].H != tensors[m.y].H || tensors[m.x].W != tensors[m.y].W || tensors[m.x].elt != 2 || tensors[m.y].elt != 2) fail("MLP tensor shape");
                stats_of(m.stats_out, m.y);
                if (elt != 2) fail("fused MLP in an fp32 plan");
                blob(m.w1, false, (size_t)2 * m.C * m.C * 2); blob(m.b1, false, (size_t)2 * m.C * 4);
                blob(m.w2, false, (size_t)2 * m.C * m.C * 2); blob(m.b2, false, (size_t)m.C * 4);
                break;
            }
            case OP_SWINATTN: {
                const SwinAttnOp& a = op.sa;
                ten(a.x, false); ten(a.y, false); ten(a.stats_out, true);
                if (a.C <= 0 || a.heads <= 0 || a.hd <= 0 || a.heads * a.hd != a.C || a.ws <= 0 || a.nwin <= 0 || a.H <= 0 || a.W <= 0) fail("window attention shape");
                if (tensors[a.x].C != a.C || tensors[a.y].C != a.C || a.H * a.W != a.nwin * a.ws * a.ws || tensors[a.x].H * tensors[a.x].W != a.H * a.W) fail("window attention geometry");
                if (a.ry >= a.H || a.rx >= a.W) fail("window attention shift");
                if (tensors[a.y].H * tensors[a.y].W != a.H * a.W || tensors[a.x].elt != 2 || tensors[a.y].elt != 2) fail("window attention tensor shape");
                stats_of(a.stats_out, a.y);
                if (elt != 2) fail("fused window attention in an fp32 plan");
                blob(a.table, false, (size_t)a.H * a.W * 4);
                blob(a.wqkv, false, (size_t)3 * a.C * a.C * 2); blob(a.bqkv, false, (size_t)3 * a.C * 4);
                blob(a.wproj, false, (size_t)a.C * a.C * 2); blob(a.bproj, false, (size_t)a.C * 4);
                blob(a.maskid, false, (size_t)a.nwin * 4);
                int nmask = 0;
                for (int i = 0; i < a.nwin; ++i) { int m; memcpy(&m, blobs[a.maskid].data.data() + 4 * (size_t)i, 4); if (m < 0) fail("window mask id"); nmask = std::max(nmask, m + 1); }
                blob(a.bias, false, (size_t)nmask * a.heads * 3 * 576 * 4);
                for (size_t i = 0; i < (size_t)a.H * a.W; ++i) { int v; memcpy(&v, blobs[a.table].data.data() + 4 * i, 4); if (v < 0 || v >= a.H * a.W) fail("window table entry"); }
                break;
            }
            default: fail("op kind");
        }
    }
}

}  // namespace w2x
