// Which shapes the kernels take - pure host predicates, shared by the lowering pass (lower.cpp decides at build time what fuses and
// what fails naming the node) and the launchers.  No HIP types here: lower.cpp and everything else that parses files builds with a plain
// host compiler for the sanitizer target (`make asan`).
#pragma once

namespace w2x {

bool swin_attn_supported(int C, int heads, int hd, int ws);   // k_swinattn96.hip / k_swinattn192u.hip: the fused attention branch
bool mlp_supported(int C);                                     // k_mlp96q.hip / k_mlp2.hip: the fused MLP branch
// which fragment order the engine stores an MLP's weights in (fragorder.h): 32x32x16 tiles for C = 96 (k_mlp96q.hip), 16x16x32 for C = 192 (mlp2_kernel<192,2,4>);
// -DW2X_MLP192_TILE32 builds the engine for mlp2q_kernel<192,4> instead (the A/B of profiles/r6_kernels/lib_mlp192_tile16_frame_level.txt).  The same move at C = 96 -
// round 2's persistent kernel on 16x16x32 tiles (tools/ab/k_mlp96p.hip) in place of k_mlp96q.hip - loses: 7.67 against 7.385 ms per frame (1.48 against 1.20 ms of MLP
// time, and the image head needs its own launch again; profiles/r6_kernels/lib_mlp96_tile16_frame_level.txt)
#if defined(W2X_MLP192_TILE32)
inline bool mlp_frag32(int C) { return C == 96 || C == 192; }
#else
inline bool mlp_frag32(int C) { return C == 96; }
#endif
bool gemm_row_stats_supported(int N);                          // k_gemm.hip: row widths whose LayerNorm statistics the epilogue can emit
bool attn_supported(int hd, int ntok);                         // k_attn.hip: the un-fused attention core

}  // namespace w2x
