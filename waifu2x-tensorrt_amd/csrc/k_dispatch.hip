// Shape dispatch of the fused transformer operators (plan.h OP_SWINATTN / OP_MLP) onto the gfx950 kernels that serve them.
// lower.cpp asks swin_attn_supported() / mlp_supported() before it fuses a branch, so a plan never holds a fused op without a
// kernel; graphs of other widths keep the un-fused GEMM + OP_ATTN plan.
#include "kernels.h"

#include <cstdlib>

namespace w2x {

hipError_t launch_mlp2(const MlpParams& p, hipStream_t s);                // k_mlp2.hip
hipError_t launch_swin_attn96(const SwinAttnParams& p, hipStream_t s);    // k_swinattn96.hip
hipError_t launch_swin_attn192u(const SwinAttnParams& p, hipStream_t s);  // k_swinattn192u.hip: three workgroups per CU, one (window, head) unit at a time (round 4; round 3's
                                                                          // two-per-CU kernel, 0.3 % slower at frame level in 4 of 4 alternating pairs, retired: git 9576837:tools/ab/k_swinattn192_r3.hip)

hipError_t launch_swin_attn(const SwinAttnParams& p, hipStream_t s) {
    if (!p.wqkv_frag || !p.wproj_frag) return hipErrorInvalidValue;       // the kernels read fragment-major weights only
    if (p.C == 96 && p.hd == 16) return launch_swin_attn96(p, s);
    if (p.C == 192 && p.hd == 32) return launch_swin_attn192u(p, s);
    return hipErrorInvalidValue;
}

hipError_t launch_mlp(const MlpParams& p, hipStream_t s) {
    if (!p.w1_frag || !p.w2_frag || !mlp_supported(p.C)) return hipErrorInvalidValue;
    return launch_mlp2(p, s);
}

}  // namespace w2x
