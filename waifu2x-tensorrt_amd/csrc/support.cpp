#include "support.h"

namespace w2x {

bool swin_attn_supported(int C, int heads, int hd, int ws) {
    return ws == 6 && heads * hd == C && ((C == 96 && hd == 16) || (C == 192 && hd == 32));
}
bool mlp_supported(int C) { return C == 96 || C == 192; }
bool gemm_row_stats_supported(int N) { return N == 32 || N == 48 || N == 64 || N == 96 || N == 128 || N == 192; }
bool attn_supported(int hd, int ntok) {
    if (ntok != 36 && ntok != 64) return false;
    return hd == 8 || hd == 16 || hd == 24 || hd == 32 || hd == 48 || hd == 64;
}

}  // namespace w2x

// ---- switches.h
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "switches.h"

namespace w2x {

namespace {
struct Field { const char* name; int Switches::*i; bool Switches::*b; const char* env; int lo, hi; };
const Field kFields[] = {
    {"superbatch", &Switches::superbatch, nullptr, "W2X_SUPERBATCH", 0, 4096},
    {"groups", &Switches::groups, nullptr, "W2X_GROUPS", 1, 4},
    {"render_parts", &Switches::render_parts, nullptr, "W2X_RENDER_PARTS", 1, 4},
    {"no_graph", nullptr, &Switches::no_graph, "W2X_NO_GRAPH", 0, 1},
    {"no_rolling", nullptr, &Switches::no_rolling, "W2X_NO_ROLLING", 0, 1},
    {"poison", nullptr, &Switches::poison, "W2X_POISON", 0, 1},
    {"check_general", nullptr, &Switches::check_general, "W2X_CHECK_GENERAL", 0, 1},
    {"roctx", nullptr, &Switches::roctx, "W2X_ROCTX", 0, 1},
    {"no_fuse", nullptr, &Switches::no_fuse, nullptr, 0, 1},
    {"no_fuse_attn", nullptr, &Switches::no_fuse_attn, nullptr, 0, 1},
    {"no_se_fold", nullptr, &Switches::no_se_fold, nullptr, 0, 1},
    {"no_fuse_head", nullptr, &Switches::no_fuse_head, nullptr, 0, 1},
    {"no_fuse_stem", nullptr, &Switches::no_fuse_stem, nullptr, 0, 1},
    {"no_fuse_up", nullptr, &Switches::no_fuse_up, nullptr, 0, 1},
    {"no_conv3h_walk", nullptr, &Switches::no_conv3h_walk, nullptr, 0, 1},
    {"no_pixgemm", nullptr, &Switches::no_pixgemm, nullptr, 0, 1},
    {"no_conv3", nullptr, &Switches::no_conv3, nullptr, 0, 1},
    {"no_conv3h", nullptr, &Switches::no_conv3h, nullptr, 0, 1},
    {"no_conv48", nullptr, &Switches::no_conv48, nullptr, 0, 1},
    {"no_stem", nullptr, &Switches::no_stem, nullptr, 0, 1},
    {"attn_valu", nullptr, &Switches::attn_valu, nullptr, 0, 1},
};
void assign(Switches& s, const Field& f, long v) {
    v = std::min<long>(f.hi, std::max<long>(f.lo, v));
    if (f.i) s.*(f.i) = (int)v; else s.*(f.b) = v != 0;
}
}  // namespace

Switches& switches() { static Switches s; return s; }

void switches_from_env() {
    Switches& s = switches();
    const Switches d;
    for (const Field& f : kFields) {
        if (!f.env) continue;
        const char* e = getenv(f.env);
        if (f.i) s.*(f.i) = e ? (int)std::min<long>(f.hi, std::max<long>(f.lo, atol(e))) : d.*(f.i);
        else s.*(f.b) = e != nullptr;       // (a flag is on when its variable exists, whatever it holds)
    }
}

bool set_switch(const char* name, long value) {
    if (!name) return false;
    for (const Field& f : kFields) if (!strcmp(f.name, name)) { assign(switches(), f, value); return true; }
    return false;
}

std::string switches_nondefault() {
    const Switches& s = switches();
    const Switches d;
    std::string out;
    for (const Field& f : kFields) {
        const bool same = f.i ? s.*(f.i) == d.*(f.i) : s.*(f.b) == d.*(f.b);
        if (same) continue;
        if (!out.empty()) out += ' ';
        out += f.name;
        if (f.i) out += "=" + std::to_string(s.*(f.i));
    }
    return out;
}

}  // namespace w2x
