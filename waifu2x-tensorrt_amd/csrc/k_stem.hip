// First convolution of a network for gfx950: 3x3, stride 1, valid, on the 3-channel tile stored as 4 halves per pixel
// (r, g, b, 0), to 32 / 48 / 64 channels with bias and LeakyReLU.  Same result as gemm_kernel (k_gemm.hip), which remains
// the reference for it (switches.h: no_stem / W2X_CHECK_GENERAL).
//
// The launch is a pure output stream (config 3: 0.3 GB written for 8 GFLOP); the general kernel gathers its A operand
// element by element for the 4-channel input and reaches 1.5 TB/s.  Here a pixel IS an operand register pair:
//   * K = 9 taps x 4 stored channels = 36, taken as three k-steps of v_mfma_f32_16x16x16_f16: k-step s covers the taps
//     4s .. 4s+3, so lane (fr, g) of the B operand holds tap 4s+g of pixel fr - one 8-byte load, no shuffling;
//   * the product is computed transposed (out^T = W X^T: rows = channels, columns = 16 pixels of an output row), and the rows
//     of the A operand are a permutation of the channels (row 4g + j of n-tile nt = channel 4 NT g + 4 nt + j), so a lane ends
//     up with 4 NT CONSECUTIVE channels of one pixel and a wave's store covers 16 whole pixels back to back (16-byte stores for
//     32 / 64 channels; with 8-byte stores at a 2 N byte stride the launch ran at 2.2 TB/s); the weights (A operand, 9 register
//     pairs for 48 channels) and the bias (initial accumulator) stay in registers for the whole workgroup;
//   * a workgroup = one output row of one tile, its four waves take the 16-pixel groups round-robin.  No LDS, no barrier.
#include "kernels.h"
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void stem_kernel(const GemmParams p, int Ho, int Wo) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / Ho, oy = blockIdx.x - b * Ho;
    const half4 zero4h = {};

    // weights [N][Kw], k = tap * 4 + channel: fragment (nt, s) of lane (channel row fr, g) = W[16 nt + fr][tap 4s+g][0..3]
    const _Float16* __restrict__ Wt = (const _Float16*)p.wt;
    half4 wf[NT][3];
    float4v bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + g;
            wf[nt][s] = tap < 9 ? *(const half4*)(Wt + (size_t)(4 * NT * (fr >> 2) + 4 * nt + (fr & 3)) * p.Kw + tap * 4) : zero4h;
        }
        bias[nt] = *(const float4v*)(p.bias + 4 * NT * g + 4 * nt);    // accumulator row 4g + j of n-tile nt = channel 4 NT g + 4 nt + j
    }
    // tap 4s+g of this lane as an element offset from the pixel under the kernel's top-left corner
    int toff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = 4 * s + g < 9 ? 4 * s + g : 8;
        toff[s] = ((tap / 3) * p.a.Ws + tap % 3) * 4;
    }
    const _Float16* __restrict__ Arow = (const _Float16*)p.a.p + ((size_t)(b * p.a.Hs + p.a.y0 + oy) * p.a.Ws + p.a.x0) * 4;
    _Float16* __restrict__ Orow = (_Float16*)p.out.p + ((size_t)(b * p.out.Hs + oy) * p.out.Ws) * p.out.Cs;
    const int ngroups = (Wo + 15) >> 4;
    for (int gi = wv; gi < ngroups; gi += 4) {
        const int px = gi * 16 + fr;
        const int pxc = px < Wo ? px : Wo - 1;                          // lanes past the row end read its last pixel, store nothing
        half4 xf[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const half4 v = *(const half4*)(Arow + (size_t)pxc * 4 + toff[s]);
            xf[s] = 4 * s + g < 9 ? v : zero4h;
        }
        half4 h[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float4v acc = bias[nt];
#pragma unroll
            for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(wf[nt][s], xf[s], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[j];
                if (p.act == 1) v = v > 0.f ? v : v * p.alpha;
                h[nt][j] = (_Float16)v;
            }
        }
        if (px < Wo) {   // the lane's 4 NT channels are consecutive: the four lanes of a pixel write its N channels back to back
            _Float16* op_ = Orow + (size_t)px * p.out.Cs + 4 * NT * g;
            if (NT == 3) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) *(half4*)(op_ + 4 * nt) = h[nt];
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; nt += 2)
                    *(half8*)(op_ + 4 * nt) = (half8){h[nt][0], h[nt][1], h[nt][2], h[nt][3], h[nt + 1][0], h[nt + 1][1], h[nt + 1][2], h[nt + 1][3]};
            }
        }
    }
}

}  // namespace

bool stem_supported(const GemmParams& p) {
    if (switches().no_stem || p.a_scale || p.res_scale || p.amode != 2 || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.omode != 0 || p.ln || (p.act != 0 && p.act != 1) || p.has_clip ||
        p.stats_out || p.pool_out || p.res.p || p.res2.p) return false;
    if (p.a.Cs != 4 || p.K != 36 || p.Kw < 36 || p.Kw % 4 || (p.N != 32 && p.N != 48 && p.N != 64) || p.out.Cs != p.N || p.aW <= 0 || p.Mrows % p.aW) return false;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    if (p.a.y0 < 0 || p.a.x0 < 0 || p.a.y0 + Ho + 2 > p.a.Hs || p.a.x0 + Wo + 2 > p.a.Ws) return false;
    return p.out.Hs >= Ho && p.out.Ws >= Wo && p.B > 0;   // rows land at (b, m / aW, m % aW) of the output view, as in gemm_kernel
}

hipError_t launch_stem(const GemmParams& p, hipStream_t s) {
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    const dim3 grid((unsigned)(p.B * Ho));
    if (p.N == 32) hipLaunchKernelGGL(stem_kernel<2>, grid, dim3(256), 0, s, p, Ho, Wo);
    else if (p.N == 48) hipLaunchKernelGGL(stem_kernel<3>, grid, dim3(256), 0, s, p, Ho, Wo);
    else hipLaunchKernelGGL(stem_kernel<4>, grid, dim3(256), 0, s, p, Ho, Wo);
    return hipGetLastError();
}

}  // namespace w2x
