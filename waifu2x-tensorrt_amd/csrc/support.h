// Which shapes the kernels take - pure host predicates, shared by the lowering pass (lower.cpp decides at build time what fuses and
// what fails naming the node) and the launchers.  No HIP types here: lower.cpp and everything else that parses files builds with a plain
// host compiler for the sanitizer target (`make asan`).
#pragma once

namespace w2x {

bool swin_attn_supported(int C, int heads, int hd, int ws);   // k_swinattn96.hip / k_swinattn192u.hip: the fused attention branch
bool mlp_supported(int C);                                     // k_mlp96q.hip / k_mlp2.hip: the fused MLP branch
bool gemm_row_stats_supported(int N);                          // k_gemm.hip: row widths whose LayerNorm statistics the epilogue can emit
bool attn_supported(int hd, int ntok);                         // k_attn.hip: the un-fused attention core

}  // namespace w2x
