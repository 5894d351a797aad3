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
