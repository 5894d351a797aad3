// swin_unet's patch stage - stem (3x3, the 4-halves-per-pixel tile -> 48 channels, LeakyReLU) + patch convolution (3x3, 48 -> 96,
// LeakyReLU / none) - as ONE PERSISTENT kernel for gfx950: the weight-resident variant of k_conv48.hip's conv48_kernel<true> (late round 4).
//
// conv48_kernel<true> runs at 37 % of the matrix pipe: a workgroup of four waves lives through halo stage -> seven weight stages with a
// barrier each -> epilogue, two workgroups per CU (44 KB halo + 24 KB of weight stages), and every workgroup streams the same 84 KB of
// weights through LDS again.  Here ONE workgroup of eight waves per CU keeps the WHOLE weight matrix in LDS for its lifetime
// (14 k-steps x 6 n-tiles of 1 KiB fragments = 84 KB, copied once, fragment-major as the engine stores them: a wave's operand is one
// conflict-free ds_read_b128) and walks over output tiles of 8 rows x 64 columns:
//     stem of the tile's 10 x 66 halo pixels from operands that were requested during the PREVIOUS tile's epilogue (42 groups of 16 pixels
//       over eight waves, stem_kernel<3>'s instruction sequence, 8-byte LDS stores)                                        | barrier
//     14 k-steps of 4 pixel-fragment reads + 6 x (1 weight-fragment read + 4 products): no barrier, no global load on the way | barrier
//     request the next tile's stem operands; epilogue through the wave's LDS tile (over the halo rows, as in k_conv48.hip),
//       16-byte row stores                                                                                                  | barrier
// Per tile three barriers instead of nine per half-size tile, no weight traffic after the first 84 KB, and the only global loads a wave
// waits for are its first tile's.  LDS: 73.9 KB halo (112-byte pixels: the 16 pixels of a fragment on different banks) + 84 KB = 159.9 KB.
// The products of an output are the same instructions in the same order as conv48_kernel's (bias as the initial accumulator, k-steps 0..13),
// so the bytes are those of the two-launch path (tests/test_gpu_parity.py::test_stem_folded_into_the_patch_convolution_is_bit_identical).
//
// MEASURED AND NOT ADOPTED (profiles/r4_kernels/r4late_conv48p_*.txt): 0.50 ms per pass against 0.32 for conv48_kernel<true>, frame +0.2 ms.  The phase
// experiments (W2X_C48P_EXP, tools/ab/c48p_phases.sh) say why: products + epilogue alone take 0.24 ms (the products ~0.17: 60 % of the matrix pipe, as
// designed), the stem phase 0.09 - and the two together 0.50, because vmcnt is ONE in-order counter for loads and stores on gfx9: the stem's wait for its
// operands (requested before the epilogue's stores) is a wait for vmcnt(0) - the compiler cannot prove how many of the predicated stores were issued - and so
// every tile drains the previous tile's 96 KB of stores with all eight waves idle (one store in sixteen: 0.32 ms; no stem: the stores drain under the next
// tile's products and cost nothing).  Hand-counted waits (operand loads as inline asm, unconditional buffer stores, biases from LDS) would take it to
// ~0.30, no better than the kernel that ships: in this lockstep design the stem phase cannot run under another tile's products (no LDS for a second halo
// tile beside 84 KB of weights).  Kept as the record of that, behind W2X_CONV48_PERSIST=1; bit-identical to the shipped path by test.
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define W2X_PHASE_FENCE() asm volatile("" ::: "memory")
#ifndef W2X_C48P_EXP
#define W2X_C48P_EXP 0      // timing experiments (wrong results): bit 0 no stem products / halo stores, bit 1 no products, bit 2 no epilogue, bit 3 no operand requests, bit 4 one store in sixteen
#endif

constexpr int CIN = 48, N = 96, NT = N / 16, SNT = CIN / 16;
constexpr int NWV = 8, TH = NWV, TW = 64, HR = TH + 2, HC = TW + 2;
constexpr int LDP = CIN + 8;                                      // halo pixel stride (halves)
constexpr int KSTEPS = 14;                                        // k-steps of 32 over the padded K = 448
constexpr int HALO = HR * HC * LDP * 2, WLDS = KSTEPS * NT * 1024;
constexpr int OT = 16 * (N + 8) * 2;                              // per-wave output m-tile
constexpr int SMEM48P = HALO + WLDS;
constexpr int NPIX = HR * HC, NGRP = (NPIX + 15) / 16, GPW = (NGRP + NWV - 1) / NWV;
static_assert(NWV * OT <= HALO && SMEM48P <= 160 * 1024, "layout");

__global__ __launch_bounds__(NWV * 64, 1) void conv48p_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y, int ntiles, const GemmParams ps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Hl = (_Float16*)smem;                                   // [HR][HC][LDP]
    _Float16* Wl = (_Float16*)(smem + HALO);                          // [NT][KSTEPS][64][8]: the engine's fragment-major copy, whole
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* Ot = (_Float16*)(smem + wv * OT);                       // [16][N+8], over the halo tile between the products and the next stem

    // ---- the weights, once
    {
        const half8* __restrict__ src = (const half8*)p.wt_frag;
        constexpr int NPIECE = WLDS / 16;
        for (int i = tid; i < NPIECE; i += NWV * 64) *(half8*)(Wl + (size_t)i * 8) = src[i];
    }
    // ---- stem constants (k_stem.hip stem_kernel<3>): lane (fr, g) of k-step t3 holds tap 4 t3 + g of output channel 12 (fr >> 2) + 4 nt + (fr & 3)
    const int Hs_o = ps.Mrows / ps.aW, Ws_o = ps.aW;                  // extent of the stem's output = this convolution's input map
    const _Float16* __restrict__ Wt = (const _Float16*)ps.wt;
    const half4 zero4h = {};
    half4 wf[SNT][3];
#pragma unroll
    for (int nt = 0; nt < SNT; ++nt)
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            const int tap = 4 * t3 + g;
            wf[nt][t3] = tap < 9 ? *(const half4*)(Wt + (size_t)(4 * SNT * (fr >> 2) + 4 * nt + (fr & 3)) * ps.Kw + tap * 4) : zero4h;
        }
    const float* __restrict__ sbias = ps.bias + 4 * SNT * g;           // accumulator row 4g + j of n-tile nt = channel 12 g + 4 nt + j (re-read per tile: registers are short here)
    int toff[3];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const int tap = 4 * t3 + g < 9 ? 4 * t3 + g : 8;
        toff[t3] = ((tap / 3) * ps.a.Ws + tap % 3) * 4;
    }
    const _Float16* __restrict__ In = (const _Float16*)ps.a.p;
    const int tpi = tiles_x * tiles_y;
    // the stem operands of tile t: three 8-byte pieces per halo pixel group of this wave (pixels beyond the map are clamped: they feed outputs nobody stores)
    half4 xf[GPW][3];
    auto request = [&](int t) {
        const int b = t / tpi, trem = t - b * tpi, ty = trem / tiles_x, tx = trem - ty * tiles_x;
#pragma unroll
        for (int k = 0; k < GPW; ++k) {
            const int pi = min((wv + NWV * k) * 16 + fr, NPIX - 1), hr = pi / HC, hc = pi - hr * HC;
            const int Y = min(p.a.y0 + ty * TH + hr, Hs_o - 1), X = min(p.a.x0 + tx * TW + hc, Ws_o - 1);
            const _Float16* src = In + ((size_t)(b * ps.a.Hs + ps.a.y0 + Y) * ps.a.Ws + ps.a.x0 + X) * 4;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const half4 v = *(const half4*)(src + toff[t3]);
                xf[k][t3] = 4 * t3 + g < 9 ? v : zero4h;
            }
        }
    };
    // this lane's piece of the A fragment of every k-step (k_conv48.hip): k = 32 s + 8 g = (tap q / 6, channels 8 (q % 6) .. +7) with q = 4 s + g
    const _Float16* arow = Hl + (wv * HC + fr) * LDP;
    auto aoff = [&](int s) {                                          // (computed per k-step: fourteen offsets would be fourteen registers this kernel does not have)
        const int q = 4 * s + g, tap = q < 54 ? q / 6 : 8, ch = q < 54 ? q - (q / 6) * 6 : 0;
        return ((tap / 3) * HC + tap % 3) * LDP + ch * 8;
    };
    const _Float16* wl = Wl + lane * 8;

    int t = blockIdx.x;
    if (t < ntiles) request(t);
    __syncthreads();                                                  // the weights are in place
    for (; t < ntiles; t += gridDim.x) {
        const int b = t / tpi, trem = t - b * tpi, ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        // ---- stem -> halo tile
#pragma unroll
        for (int k = 0; k < GPW; ++k) {
            const int gi = wv + NWV * k, pi = gi * 16 + fr;
            if (gi >= NGRP || (W2X_C48P_EXP & 1)) break;
#pragma unroll
            for (int nt = 0; nt < SNT; ++nt) {
                float4v a4 = *(const float4v*)(sbias + 4 * nt);
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) a4 = __builtin_amdgcn_mfma_f32_16x16x16f16(wf[nt][t3], xf[k][t3], a4, 0, 0, 0);
                half4 hq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = a4[j];
                    if (ps.act == 1) v = v > 0.f ? v : v * ps.alpha;
                    hq[j] = (_Float16)v;
                }
                if (pi < NPIX) *(half4*)(Hl + pi * LDP + 4 * SNT * g + 4 * nt) = hq;
            }
        }
        __syncthreads();
        // ---- products: bias as the initial accumulator, the k-steps in order
        float4v acc[4][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4v bv = *(const float4v*)(p.bias + nt * 16 + g * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = bv;
        }
#pragma unroll
        for (int s = 0; s < ((W2X_C48P_EXP & 2) ? 0 : KSTEPS); ++s) {
            half8 xa[4];
            const _Float16* ap = arow + aoff(s);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xa[mt] = *(const half8*)(ap + mt * 16 * LDP);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const half8 wb = *(const half8*)(wl + (size_t)(nt * KSTEPS + s) * 512);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, xa[mt], acc[mt][nt], 0, 0, 0);
            }
            W2X_PHASE_FENCE();                                        // operand reads stay with their k-step
        }
        __syncthreads();                                              // every wave is done with the halo tile
        if (t + (int)gridDim.x < ntiles && !(W2X_C48P_EXP & 8)) request(t + gridDim.x);      // in flight under this tile's epilogue (under the products they would be 36 registers the product loop does not have)
        W2X_PHASE_FENCE();
        // ---- epilogue: output row oy0 + wv, four m-tiles of 16 pixels through the wave's LDS tile
        const int oy = oy0 + wv;
        _Float16* __restrict__ Og = (_Float16*)p.out.p + ((size_t)(b * p.out.Hs + oy) * p.out.Ws + ox0) * p.out.Cs;
        constexpr int PPO = N / 8, NPO = 16 * PPO / 64;
#pragma unroll
        for (int mt = 0; mt < ((W2X_C48P_EXP & 4) ? 0 : 4); ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float4v v = acc[mt][nt];
                if (p.act == 1) v = __builtin_elementwise_max(v, v * p.alpha);      // LeakyReLU with a slope in [0, 1] (conv48_supported) = max(v, slope v)
                *(half4*)(Ot + fr * (N + 8) + nt * 16 + g * 4) = (half4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
            W2X_PHASE_FENCE();
#pragma unroll
            for (int k = 0; k < NPO; ++k) {
                const int idx = k * 64 + lane, px = idx / PPO, c = idx - px * PPO;
                if (oy < Ho && ox0 + mt * 16 + px < Wo && !((W2X_C48P_EXP & 16) && px != 15))
                    *(half8*)(Og + (size_t)(mt * 16 + px) * p.out.Cs + c * 8) = *(const half8*)(Ot + px * (N + 8) + c * 8);
            }
            W2X_PHASE_FENCE();
        }
        __syncthreads();                                              // the waves' output tiles lie over halo rows the next stem writes
    }
}

}  // namespace

bool conv48p_enabled() { return getenv("W2X_CONV48_PERSIST") != nullptr && getenv("W2X_CONV48_PERSIST")[0] == '1'; }   // off unless asked for (see the header); read per launch of an eager pass, baked into captured graphs

// p, ps: a pair conv48_stem_supported() (k_conv48.hip) accepts
hipError_t launch_conv48p(const GemmParams& p, const GemmParams& ps, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv48p_kernel, SMEM48P, lds_ok); e != hipSuccess) return e;
    static int cus[64] = {};      // CUs per device ordinal (one workgroup each)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!cus[dev]) { int n = 0; if (hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e; cus[dev] = std::max(1, n); }
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH, ntiles = p.B * tiles_x * tiles_y;
    hipLaunchKernelGGL(conv48p_kernel, dim3((unsigned)std::min(ntiles, cus[dev])), dim3(NWV * 64), SMEM48P, s, p, Ho, Wo, tiles_x, tiles_y, ntiles, ps);
    return hipGetLastError();
}

}  // namespace w2x
