#include "plan.h"

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
      << "] ops=" << ops.size() << " tensors=" << tensors.size() << " blobs=" << blobs.size() << " flops=" << (long long)flops << "\n";
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
constexpr uint32_t kVersion = 8;
}  // namespace

std::vector<uint8_t> Plan::serialize() const {
    Writer w;
    w.pod(kMagic); w.pod(kVersion);
    w.pod<uint32_t>(sizeof(GemmOp)); w.pod<uint32_t>(sizeof(AttnOp)); w.pod<uint32_t>(sizeof(SeOp) + sizeof(MlpOp) + sizeof(SwinAttnOp));
    w.pod(B); w.pod(userB); w.pod(Cin); w.pod(T); w.pod(Tout); w.pod(Cout); w.pod(in_tensor); w.pod(out_tensor); w.pod(flops);
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
    pl.B = r.pod<int>(); pl.userB = r.pod<int>(); pl.Cin = r.pod<int>(); pl.T = r.pod<int>(); pl.Tout = r.pod<int>(); pl.Cout = r.pod<int>();
    pl.in_tensor = r.pod<int>(); pl.out_tensor = r.pod<int>(); pl.flops = r.pod<double>();
    pl.model_kind = r.str();
    uint64_t nt = r.pod<uint64_t>(); pl.tensors.resize(nt); for (auto& t : pl.tensors) t = r.pod<TensorDesc>();
    uint64_t nb = r.pod<uint64_t>(); pl.blobs.resize(nb); for (auto& b : pl.blobs) b.data = r.bytes();
    uint64_t no = r.pod<uint64_t>(); pl.ops.resize(no);
    for (auto& op : pl.ops) { op.kind = r.pod<int>(); op.name = r.str(); op.g = r.pod<GemmOp>(); op.at = r.pod<AttnOp>(); op.se = r.pod<SeOp>(); op.m = r.pod<MlpOp>(); op.sa = r.pod<SwinAttnOp>(); op.flops = r.pod<double>(); }
    return pl;
}

}  // namespace w2x
