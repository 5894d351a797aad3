// Direct 3x3 convolution (stride 1, valid) from 48 to 96 channels for gfx950 - swin_unet's patch convolution, the last launch of
// that graph that ran on the general implicit-GEMM kernel (k_gemm.hip: every input pixel fetched nine times from L2, 1.64x the
// algorithmic HBM bytes, 19 % MFMA busy).  Same tiling as k_conv3.hip - a workgroup owns 4 output rows x 64 columns, the
// 6 x 66 pixel halo tile goes to LDS once and serves all nine taps, wave w computes output row w as four 16-pixel m-tiles x six
// n-tiles - but 48 channels are not a whole number of 32-wide k-steps, so K = 9 taps x 48 channels = 432 is taken as ONE flat
// sequence of 14 k-steps (the last half step meets zero weights): lane group g of k-step s covers k = 32s + 8g .. +7, which is 8
// consecutive channels of ONE tap ((4s+g) / 6) because 48 is a multiple of 8 - its A fragment is still a single 16-byte LDS read,
// only the (tap, channel) offset differs per lane group.  The whole halo tile (all 48 channels, 44 KB) is staged once; weights
// stream through LDS two k-steps at a time (fragment-major copy padded to 448 columns, staged by the four waves, double-buffered,
// one barrier per stage).  The product is computed transposed (rows = output channels), so the epilogue is a float4 bias as the initial
// accumulator, LeakyReLU / none on 4 consecutive channels of a pixel, 8-byte stores into a per-wave LDS tile, 16-byte row stores.
#include "kernels.h"
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define W2X_PHASE_FENCE() asm volatile("" ::: "memory")

constexpr int CIN = 48, N = 96, NT = N / 16;
constexpr int TH = 4, TW = 64, HR = TH + 2, HC = TW + 2;      // output tile, halo tile (pixels)
constexpr int LDP = CIN + 8;                                    // halo pixel stride (halves): 112 B, the 16 pixels of a fragment fall on different banks
constexpr int KSTEPS = 14, SK = 2, NSTAGE = KSTEPS / SK;        // k-steps of 32 over the padded K = 448; per weight stage
constexpr int NF = SK * NT, NFW = NF / 4;                       // weight fragments (KiB) per stage / per wave
constexpr int HALO = HR * HC * LDP * 2, WBUF = NF * 1024;
constexpr int OT = 16 * (N + 8) * 2;                            // per-wave output m-tile
constexpr int SMEM48 = HALO + 2 * WBUF;
constexpr int PPP = CIN / 8;                                    // 16-byte pieces per halo pixel
static_assert(4 * OT <= HALO && NF % 4 == 0 && KSTEPS % SK == 0, "layout");

// STEM (round 4): the launch also computes its own input.  swin_unet's patch stage is stem (3x3, the 4-halves-per-pixel tile -> 48 channels, k_stem.hip) followed by this
// convolution, and the 48-channel map between them (0.28 GB per pass at config 3) is written by one launch and read by the next and by nothing else.  With ps = the
// stem's parameters the halo tile is not fetched but COMPUTED: the 6 x 66 halo pixels are 25 groups of 16, the four waves take them round-robin, each group is
// stem_kernel's three 16x16x16 k-steps on operands read straight from the input tile (L2-resident: 24 MB per pass), bias as the initial accumulator, LeakyReLU, fp16 -
// the same instructions on the same operands as stem_kernel, so the halo tile holds the bytes that kernel would have stored (bit-identical frames by test) - and goes to
// LDS in 8-byte pieces (a lane ends with 12 consecutive channels of one pixel).  The stem's products are recomputed for the halo (x 1.55), 17 % on top of this launch's
// matrix work; the stem launch (0.07 ms), its 0.28 GB of stores and this launch's 0.28 GB of halo loads go.
template <bool STEM>
__global__ __launch_bounds__(256, 2) void conv48_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y, const GemmParams ps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Hl = (_Float16*)smem;                                   // [HR][HC][LDP]
    _Float16* WB = (_Float16*)(smem + HALO);                          // [2][NF][64][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* Ot = (_Float16*)(smem + wv * OT);                       // [16][N+8], over the halo tile once the last tap is done (the loop ends with a barrier)

    const int tpi = tiles_x * tiles_y;
    const int b = blockIdx.x / tpi, trem = blockIdx.x - b * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const _Float16* __restrict__ Wf = (const _Float16*)p.wt_frag + lane * 8;        // [NT][KSTEPS][64][8]
    auto frag_src = [&](int sg, int f) { const int sl = f / NT, nt = f - sl * NT; return Wf + (size_t)(nt * KSTEPS + sg * SK + sl) * 512; };   // fragment f = (local k-step, n-tile) of stage sg

    half8 stg[NFW];
#pragma unroll
    for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(0, wv * NFW + i);

    const int hrows = min(HR, Ho + 2 - oy0), hcols = min(HC, Wo + 2 - ox0);
    if (STEM) {
        // ---- halo tile computed from the network's input tile (see the kernel's header): stem_kernel<3>'s arithmetic per group of 16 halo pixels
        constexpr int SNT = 3, NPIX = HR * HC, NGRP = (NPIX + 15) / 16, GPW = (NGRP + 3) / 4;
        const int Hs_o = ps.Mrows / ps.aW, Ws_o = ps.aW;                    // extent of the stem's output = this convolution's input map
        const _Float16* __restrict__ Wt = (const _Float16*)ps.wt;
        const half4 zero4h = {};
        half4 wf[SNT][3];
        float4v sbias[SNT];
#pragma unroll
        for (int nt = 0; nt < SNT; ++nt) {
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const int tap = 4 * t3 + g;
                wf[nt][t3] = tap < 9 ? *(const half4*)(Wt + (size_t)(4 * SNT * (fr >> 2) + 4 * nt + (fr & 3)) * ps.Kw + tap * 4) : zero4h;
            }
            sbias[nt] = *(const float4v*)(ps.bias + 4 * SNT * g + 4 * nt);  // accumulator row 4g + j of n-tile nt = channel 12 g + 4 nt + j
        }
        int toff[3];
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
            const int tap = 4 * t3 + g < 9 ? 4 * t3 + g : 8;
            toff[t3] = ((tap / 3) * ps.a.Ws + tap % 3) * 4;
        }
        const _Float16* __restrict__ In = (const _Float16*)ps.a.p;
        half4 xf[GPW][3];
#pragma unroll
        for (int k = 0; k < GPW; ++k) {                                     // every operand of the wave's groups is requested before the first product
            const int pi = min((wv + 4 * k) * 16 + fr, NPIX - 1), hr = pi / HC, hc = pi - hr * HC;
            const int Y = min(p.a.y0 + oy0 + hr, Hs_o - 1), X = min(p.a.x0 + ox0 + hc, Ws_o - 1);   // (halo pixels beyond the map feed outputs nobody stores)
            const _Float16* src = In + ((size_t)(b * ps.a.Hs + ps.a.y0 + Y) * ps.a.Ws + ps.a.x0 + X) * 4;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const half4 v = *(const half4*)(src + toff[t3]);
                xf[k][t3] = 4 * t3 + g < 9 ? v : zero4h;
            }
        }
#pragma unroll
        for (int k = 0; k < GPW; ++k) {
            const int gi = wv + 4 * k, pi = gi * 16 + fr;
            if (gi >= NGRP) break;
#pragma unroll
            for (int nt = 0; nt < SNT; ++nt) {
                float4v a4 = sbias[nt];
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
    } else {
    // ---- halo tile, all 48 channels (valid convolution: input extent = output extent + 2; pixels beyond it are zeros).  Every piece of the
    // tile is requested before the first one is stored (round 2's loop fetched, waited and stored piece by piece: ten memory round trips in a
    // row per workgroup - most of its life); pieces outside the map go through the buffer resource's bounds check (offset 0xFFFFFFFF reads zeros).
    {
        constexpr int NPIECE = HR * HC * PPP, NIT = (NPIECE + 255) / 256;
        const size_t map_bytes = (size_t)p.B * p.a.Hs * p.a.Ws * CIN * 2;          // conv48_supported() keeps this below 4 GB
        const __amdgpu_buffer_rsrc_t AB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a.p), 0, (unsigned)map_bytes, 0x00020000);
        const unsigned base = (unsigned)((((size_t)(b * p.a.Hs + p.a.y0 + oy0) * p.a.Ws + p.a.x0 + ox0) * CIN) * 2);
        half8 hv[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = k * 256 + tid, pix = i / PPP, c8 = i - pix * PPP, hr = pix / HC, hc = pix - hr * HC;
            const unsigned off = (i < NPIECE && hr < hrows && hc < hcols) ? base + (unsigned)((hr * p.a.Ws + hc) * CIN + c8 * 8) * 2u : 0xFFFFFFFFu;
            hv[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(AB, off, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = k * 256 + tid, pix = i / PPP, c8 = i - pix * PPP;
            if (i < NPIECE) *(half8*)(Hl + pix * LDP + c8 * 8) = hv[k];
        }
    }
    }
#pragma unroll
    for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];

    // the product is computed transposed (rows = output channels, columns = 16 pixels), so a lane ends with 4 consecutive channels
    // of one pixel: the bias is a float4 initial accumulator and the tile reaches LDS in 8-byte stores
    float4v acc[4][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float4v bv = *(const float4v*)(p.bias + nt * 16 + g * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = bv;
    }
    // this lane's piece of the A fragment of every k-step: k = 32s + 8g = (tap q / 6, channels 8 (q % 6) .. +7) with q = 4s + g;
    // q >= 54 (the padded tail) reads tap 8 again and meets zero weights
    const _Float16* arow = Hl + (wv * HC + fr) * LDP;
    int aoff[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int q = 4 * s + g, tap = q < 54 ? q / 6 : 8, ch = q < 54 ? q - (q / 6) * 6 : 0;
        aoff[s] = ((tap / 3) * HC + tap % 3) * LDP + ch * 8;
    }
    __syncthreads();

#pragma unroll
    for (int sg = 0; sg < NSTAGE; ++sg) {
        const _Float16* wcur = WB + (size_t)(sg & 1) * (WBUF / 2) + lane * 8;
        if (sg + 1 < NSTAGE) {
#pragma unroll
            for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(sg + 1, wv * NFW + i);
            W2X_PHASE_FENCE();     // the requests stay HERE, under this stage's products (the scheduler sank them to the LDS stores at its end: an L2
        }                          // round trip exposed in front of every barrier)
#pragma unroll
        for (int sl = 0; sl < SK; ++sl) {
            const int s = sg * SK + sl;
            half8 xa[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xa[mt] = *(const half8*)(arow + mt * 16 * LDP + aoff[s]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const half8 wb = *(const half8*)(wcur + (size_t)(sl * NT + nt) * 512);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, xa[mt], acc[mt][nt], 0, 0, 0);
            }
        }
        if (sg + 1 < NSTAGE) {
#pragma unroll
            for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)((sg + 1) & 1) * (WBUF / 2) + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];
        }
        __syncthreads();
    }

    // ---- epilogue: output row oy0 + wv, four m-tiles of 16 pixels through the wave's LDS tile
    const int oy = oy0 + wv;
    _Float16* __restrict__ Og = (_Float16*)p.out.p + ((size_t)(b * p.out.Hs + oy) * p.out.Ws + ox0) * p.out.Cs;
    constexpr int PPO = N / 8, NPO = 16 * PPO / 64;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
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
            if (oy < Ho && ox0 + mt * 16 + px < Wo)
                w2x_store_out((half8*)(Og + (size_t)(mt * 16 + px) * p.out.Cs + c * 8), *(const half8*)(Ot + px * (N + 8) + c * 8));
        }
        W2X_PHASE_FENCE();
    }
}

}  // namespace

bool conv48_supported(const GemmParams& p) {
    if (switches().no_conv48 || p.a_scale || !p.wt_frag || p.amode != 2 || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.omode != 0 || p.ln || (p.act != 0 && p.act != 1) ||
        p.has_clip || p.stats_out || p.pool_out || p.res.p || p.res2.p) return false;
    if (p.act == 1 && !(p.alpha >= 0.f && p.alpha <= 1.f)) return false;
    if (p.a.Cs != CIN || p.N != N || p.K != 9 * CIN || p.out.Cs != N || p.aW <= 0 || p.Mrows % p.aW) return false;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    if ((size_t)p.B * p.a.Hs * p.a.Ws * CIN * 2 >= ((size_t)1 << 32) - 65536) return false;   // the halo tile is fetched through a 32-bit buffer resource
    return p.a.y0 + Ho + 2 <= p.a.Hs && p.a.x0 + Wo + 2 <= p.a.Ws && p.out.Hs >= Ho && p.out.Ws >= Wo;
}

hipError_t launch_conv48(const GemmParams& p, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv48_kernel<false>, SMEM48, lds_ok); e != hipSuccess) return e;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    hipLaunchKernelGGL(conv48_kernel<false>, dim3((unsigned)(p.B * tiles_x * tiles_y)), dim3(256), SMEM48, s, p, Ho, Wo, tiles_x, tiles_y, GemmParams{});
    return hipGetLastError();
}

// the stem (ps: a launch stem_supported() takes, 48 output channels) folded into the patch convolution that is the only reader of its output (p.a = a view of ps.out)
bool conv48_stem_supported(const GemmParams& p, const GemmParams& ps) {
    if (switches().no_fuse_stem || !conv48_supported(p) || !stem_supported(ps) || ps.N != CIN || ps.out.Cs != CIN || p.a.p != ps.out.p || ps.out.y0 || ps.out.x0) return false;
    const int Hs_o = ps.Mrows / ps.aW, Ws_o = ps.aW, Ho = p.Mrows / p.aW, Wo = p.aW;
    return p.a.Hs == ps.out.Hs && p.a.Ws == ps.out.Ws && p.a.y0 + Ho + 2 <= Hs_o && p.a.x0 + Wo + 2 <= Ws_o && p.B == ps.B;
}

hipError_t launch_conv48_stem(const GemmParams& p, const GemmParams& ps, hipStream_t s) {
    static unsigned lds_ok = 0;
    if (hipError_t e = ensure_dynamic_lds((const void*)conv48_kernel<true>, SMEM48, lds_ok); e != hipSuccess) return e;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    hipLaunchKernelGGL(conv48_kernel<true>, dim3((unsigned)(p.B * tiles_x * tiles_y)), dim3(256), SMEM48, s, p, Ho, Wo, tiles_x, tiles_y, ps);
    return hipGetLastError();
}

}  // namespace w2x
