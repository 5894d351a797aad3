// A/B record (not built into libw2x.so): the first schedule of the direct 3x3 convolution, replaced by csrc/k_conv3.hip in round 2.
// One tap of one 32-channel chunk per barrier, both operands through LDS, 4 x 64 output tile: 460-620 TFLOP/s on cunet's layers.
// Direct 3x3 convolution (stride 1, valid) for gfx950 with the input staged ONCE per output tile:
//     out[b][y][x][n] = act( sum_{ky,kx,c} in[b][y+ky][x+kx][c] * W[n][(ky*3+kx)*Cin + c] + bias[n] )
// gemm_kernel (k_gemm.hip) treats the same convolution as an implicit GEMM whose A rows are gathered per K-chunk, i.e. every
// input pixel is fetched from L2 nine times (once per tap); on cunet that is 10 GB of L2 reads for a 1.2 GB activation and the
// convolutions sit at ~10 % of the MFMA peak.  Here a workgroup owns an output tile of 4 rows x 64 columns: per chunk of KC input
// channels the 6 x 66 pixel halo tile goes to LDS once and serves all nine taps (the A fragment of tap (ky,kx) is just a shifted
// 16-byte read, pixel stride KC+8 halves so the 16 pixels of a fragment fall on different banks); weights stream through LDS per
// kernel row (three taps; fragment-major copy, staged once per workgroup by its four waves, double-buffered, one barrier per stage).  Wave w computes
// output row w of the tile: four 16-pixel m-tiles x N/16 n-tiles of accumulators for the whole K loop.
// Epilogue: bias (initial accumulator), LeakyReLU / none, fp16, through a small per-wave LDS tile into 16-byte row stores.
// Covers the plain convolutions (rows output, no LayerNorm / residual / clip / statistics / pooling); the rest stays on gemm_kernel.
#include "kernels.h"
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define W2X_PHASE_FENCE() asm volatile("" ::: "memory")
// sum over the four 16-lane rows of a wave (see k_swinattn.hip for why the swaps are inline asm on two registers)
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_sum(float v) { float a = v, b = v; swap16(a, b); v = a + b; a = v; b = v; swap32(a, b); return a + b; }

template <int KC, int N, int TS_>
struct Conv3Cfg {
    static constexpr int TH = 4, TW = 64, HR = TH + 2, HC = TW + 2;   // output tile, halo tile (pixels)
    static constexpr int LDP = KC + 8;                                  // halo pixel stride (halves)
    static constexpr int KS = KC / 32, NT = N / 16;
    static constexpr int HALO = HR * HC * LDP * 2;                      // bytes
    static constexpr int TS = TS_;                                      // taps per weight stage (3 = one kernel row; 1 where LDS is short)
    static constexpr int NF = TS * NT * KS, NFW = NF / 4;               // weight fragments per stage / per wave
    static constexpr int WBUF = NF * 1024;
    static constexpr int OT = 16 * (N + 8) * 2;                         // per-wave output m-tile
    static constexpr int SMEM = HALO + 2 * WBUF;                        // the per-wave output tiles reuse the halo area after the last tap
    static_assert(4 * OT <= HALO, "output tiles alias the halo tile");
    static constexpr int PPP = KC / 8;                                  // 16-byte pieces per halo pixel
    static_assert(NF % 4 == 0, "fragments per wave");
};

template <int KC, int N, int TS>
__global__ __launch_bounds__(256, 2) void conv3_kernel(const GemmParams p, int Ho, int Wo, int tiles_x, int tiles_y, int n0) {
    using C = Conv3Cfg<KC, N, TS>;
    constexpr int TH = C::TH, TW = C::TW, HR = C::HR, HC = C::HC, LDP = C::LDP, KS = C::KS, NT = C::NT, NF = C::NF, NFW = C::NFW, PPP = C::PPP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Hl = (_Float16*)smem;                                   // [HR][HC][LDP]
    _Float16* WB = (_Float16*)(smem + C::HALO);                       // [2][NF][64][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    _Float16* Ot = (_Float16*)(smem + wv * C::OT);                    // [16][N+8], over the halo tile once the last tap is done (the loop ends with a barrier)

    const int tpi = tiles_x * tiles_y;
    const int b = blockIdx.x / tpi, trem = blockIdx.x - b * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int Cin = p.a.Cs, nchunk = Cin / KC, KST = p.K / 32;
    const _Float16* __restrict__ Ag = (const _Float16*)p.a.p + ((size_t)(b * p.a.Hs + p.a.y0 + oy0) * p.a.Ws + p.a.x0 + ox0) * Cin;
    const _Float16* __restrict__ Wf = (const _Float16*)p.wt_frag + lane * 8;        // [N_total/16][KST][64][8]
    const int nt_base = n0 / 16;
    // weight fragment f = (kx * NT + nt) * KS + ks of stage (chunk kc, kernel row ky): n-tile nt_base + nt, k-step ((ky*3+kx) * Cin + kc * KC) / 32 + ks
    constexpr int SPC = 9 / TS;                                        // stages per channel chunk
    auto frag_src = [&](int kc, int sg, int f) {                      // fragment f = (local tap * NT + nt) * KS + ks of stage sg of chunk kc
        const int tl = f / (NT * KS), r2 = f - tl * (NT * KS), nt = r2 / KS, ks = r2 - nt * KS;
        return Wf + (size_t)((nt_base + nt) * KST + ((sg * TS + tl) * Cin + kc * KC) / 32 + ks) * 512;
    };

    float4v acc[4][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float bv = p.bias[n0 + nt * 16 + fr];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = (float4v){bv, bv, bv, bv};
    }
    half8 stg[NFW];
    const int nstage = nchunk * SPC;
#pragma unroll
    for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(0, 0, wv * NFW + i);

    // rows / columns of the halo tile that exist in the input (valid convolution: input extent = output extent + 2)
    const int hrows = min(HR, Ho + 2 - oy0), hcols = min(HC, Wo + 2 - ox0);
    int stage = 0;
#pragma unroll 1
    for (int kc = 0; kc < nchunk; ++kc) {
        __syncthreads();                                   // previous chunk's taps are done with the halo tile
        for (int i = tid; i < HR * HC * PPP; i += 256) {
            const int pix = i / PPP, c8 = i - pix * PPP, hr = pix / HC, hc = pix - hr * HC;
            half8 h = {};
            if (hr < hrows && hc < hcols) h = *(const half8*)(Ag + ((size_t)hr * p.a.Ws + hc) * Cin + kc * KC + c8 * 8);
            *(half8*)(Hl + pix * LDP + c8 * 8) = h;
        }
        if (kc == 0) {
#pragma unroll
            for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];
        }
        __syncthreads();
#pragma unroll 1
        for (int sg = 0; sg < SPC; ++sg, ++stage) {
            const _Float16* wcur = WB + (size_t)(stage & 1) * (C::WBUF / 2) + lane * 8;
            if (stage + 1 < nstage) {
                const int sg1 = sg == SPC - 1 ? 0 : sg + 1, kc1 = sg == SPC - 1 ? kc + 1 : kc;
#pragma unroll
                for (int i = 0; i < NFW; ++i) stg[i] = *(const half8*)frag_src(kc1, sg1, wv * NFW + i);
            }
#pragma unroll
            for (int tl = 0; tl < TS; ++tl) {
                const int t = sg * TS + tl, ky = t / 3, kx = t - ky * 3;
                const _Float16* arow = Hl + ((wv + ky) * HC + kx + fr) * LDP + g * 8;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    half8 xa[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) xa[mt] = *(const half8*)(arow + mt * 16 * LDP + ks * 32);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const half8 wb = *(const half8*)(wcur + (size_t)((tl * NT + nt) * KS + ks) * 512);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa[mt], wb, acc[mt][nt], 0, 0, 0);
                    }
                }
            }
            if (stage + 1 < nstage) {
#pragma unroll
                for (int i = 0; i < NFW; ++i) *(half8*)(WB + (size_t)((stage + 1) & 1) * (C::WBUF / 2) + (size_t)(wv * NFW + i) * 512 + lane * 8) = stg[i];
            }
            __syncthreads();
        }
    }

    // ---- epilogue: output row oy0 + wv, four m-tiles of 16 pixels through the wave's LDS tile
    const int oy = oy0 + wv;
    _Float16* __restrict__ Og = (_Float16*)p.out.p + ((size_t)(b * p.out.Hs + oy) * p.out.Ws + ox0) * p.out.Cs + n0;
    constexpr int PPO = N / 8, NPO = 16 * PPO / 64 > 0 ? 16 * PPO / 64 : 1;
    float csum[NT];                       // squeeze-excite pooling: column sums of the stored (fp16-rounded) values of this wave's row
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) csum[nt] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[mt][nt][j];
                if (p.act == 1) v = v > 0.f ? v : v * p.alpha;
                const _Float16 h = (_Float16)v;
                Ot[(g * 4 + j) * (N + 8) + nt * 16 + fr] = h;
                if (p.pool_out && oy < Ho && ox0 + mt * 16 + g * 4 + j < Wo) csum[nt] += (float)h;
            }
        W2X_PHASE_FENCE();
#pragma unroll
        for (int k = 0; k < NPO; ++k) {
            const int idx = k * 64 + lane, px = idx / PPO, c = idx - px * PPO;
            if (idx < 16 * PPO && oy < Ho && ox0 + mt * 16 + px < Wo)
                *(half8*)(Og + (size_t)(mt * 16 + px) * p.out.Cs + c * 8) = *(const half8*)(Ot + px * (N + 8) + c * 8);
        }
        W2X_PHASE_FENCE();
    }
    if (p.pool_out) {   // per-workgroup partial sums in a fixed order (rows of a wave, then waves 0..3): se_kernel adds the tiles of an image
        float* ws = (float*)WB;                                  // [4][N]; the weight buffers are idle (last tap ended with a barrier)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { const float t = rows_sum(csum[nt]); if (g == 0) ws[wv * N + nt * 16 + fr] = t; }
        __syncthreads();
        for (int n = tid; n < N; n += 256) p.pool_out[(size_t)blockIdx.x * p.out.Cs + n0 + n] = ws[n] + ws[N + n] + ws[2 * N + n] + ws[3 * N + n];
    }
}

template <int KC, int N, int TS>
hipError_t launch_c3(const GemmParams& p, int Ho, int Wo, int n0, hipStream_t s) {
    using C = Conv3Cfg<KC, N, TS>;
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)conv3_kernel<KC, N, TS>, C::SMEM, lds_ok); e != hipSuccess) return e;
    const int tiles_x = (Wo + C::TW - 1) / C::TW, tiles_y = (Ho + C::TH - 1) / C::TH;
    hipLaunchKernelGGL((conv3_kernel<KC, N, TS>), dim3((unsigned)(p.B * tiles_x * tiles_y)), dim3(256), C::SMEM, s, p, Ho, Wo, tiles_x, tiles_y, n0);
    return hipGetLastError();
}

}  // namespace

int conv3_tiles(const GemmParams& p) { const int Ho = p.Mrows / p.aW, Wo = p.aW; return ((Wo + 63) / 64) * ((Ho + 3) / 4); }   // workgroups per image

bool conv3_supported(const GemmParams& p) {
    static const bool off = getenv("W2X_NO_CONV3") != nullptr;   // A/B switch
    if (off || !p.wt_frag || p.amode != 2 || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.omode != 0 || p.ln || (p.act != 0 && p.act != 1) ||
        p.has_clip || p.stats_out || p.res.p || p.res2.p) return false;
    const int Cin = p.a.Cs;
    if (p.K != 9 * Cin || p.Kw != p.K || Cin % 32 || p.out.Cs != p.N || p.aW <= 0 || p.Mrows % p.aW) return false;
    const int Ho = p.Mrows / p.aW, Wo = p.aW;
    if (p.a.y0 + Ho + 2 > p.a.Hs || p.a.x0 + Wo + 2 > p.a.Ws || p.out.Hs < Ho || p.out.Ws < Wo) return false;
    // pooling partials: one per workgroup; the plan sized the buffer for ceil(Mrows / kGemmBM) row tiles per image
    if (p.pool_out && conv3_tiles(p) > (p.Mrows + kGemmBM - 1) / kGemmBM) return false;
    return p.N == 64 || p.N == 128 || p.N == 256;
}

hipError_t launch_conv3(const GemmParams& p, hipStream_t s) {
    const int Ho = p.Mrows / p.aW, Wo = p.aW, Cin = p.a.Cs;
    (void)Cin;   // chunks of 32 input channels for every width: 49 KB (N = 64) / 65 KB (N = 128) of LDS, 3 / 2 workgroups per CU
    if (p.N == 64) return launch_c3<32, 64, 1>(p, Ho, Wo, 0, s);   // 3 taps per stage (fewer barriers, 2 instead of 3 workgroups per CU) measured 19 % slower
    hipError_t e = launch_c3<32, 128, 1>(p, Ho, Wo, 0, s);
    if (e == hipSuccess && p.N == 256) e = launch_c3<32, 128, 1>(p, Ho, Wo, 128, s);
    return e;
}

}  // namespace w2x
