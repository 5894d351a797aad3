// Fused Swin attention branch for gfx950 (window 6x6, C = 96 / head dim 16  or  C = 192 / head dim 32, 6 heads):
//     y = x + proj( W-MSA( LayerNorm(x) ) )      with cyclic shift + window partition as a row table
// One launch replaces LayerNormalization, the roll/partition Slice/Concat/Reshape/Transpose chain, the QKV MatMul+Add,
// the per-head scale / QK^T / bias (+mask) / Softmax / AV chain, the head merge, the proj MatMul+Add, window reverse,
// reverse roll and the residual Add of the ONNX graph (inside TensorRT's enqueueV3, img2img_infer.cpp:80).
//
// A workgroup (4 waves) owns G whole windows (G*36 token rows).  x rows are gathered through the window table,
// normalised into LDS, and per head: q,k (computed transposed so each lane holds 4 consecutive features of one
// token) and v^T land in LDS; S^T = K Q^T, softmax over the lane-local keys (+2 cross-lane steps), and O^T = V^T P^T
// run on v_mfma_f32_16x16x32_f16 with the accumulator of S^T used directly as the B operand of the second product
// (k order permuted consistently on the V^T side).  The head outputs are collected in LDS, multiplied by Wproj, and
// the result tile goes through LDS so the residual add and the scattered HBM stores are 16-byte row pieces.
#include "kernels.h"

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    if (LPR == 32) v += __shfl_xor(v, 16);
    return v;
}

template <int C, int HD>
struct SwinCfg {
    static constexpr int NTOK = 36;
    static constexpr int G = C == 96 ? 4 : 2;          // windows per workgroup
    static constexpr int R = G * NTOK;                 // token rows
    static constexpr int RT = (R + 15) / 16, RP = RT * 16;
    static constexpr int RPQ = RP + 16;                // rows incl. the overrun of the last window's third 16-row tile
    static constexpr int NH = C / HD;
    static constexpr int LDX = C + 8, LDQ = HD + 8, LDV = RPQ + 8;
    static constexpr int WROWS = (3 * HD > 96 ? 3 * HD : 96);   // weight buffer rows: a head's q,k,v rows or a 96-row proj chunk
    static constexpr int XS = RP * LDX, OS = RP * LDX, QS = RPQ * LDQ, VS = HD * LDV, WS = WROWS * LDX;
    static constexpr int SMEM = (XS + OS + 2 * QS + VS + WS) * 2;
};

template <int C, int HD>
__global__ __launch_bounds__(256, 1) void swin_attn_kernel(const SwinAttnParams p) {
    using K = SwinCfg<C, HD>;
    constexpr int NTOK = K::NTOK, G = K::G, R = K::R, RT = K::RT, RP = K::RP, RPQ = K::RPQ, NH = K::NH;
    constexpr int LDX = K::LDX, LDQ = K::LDQ, LDV = K::LDV;
    constexpr int LPR = C == 96 ? 16 : 32, PPR = C / 8, RPP = 256 / LPR;
    constexpr int NPASS = (RP + RPP - 1) / RPP;
    constexpr int NTQ = 3 * HD / 16, NTH = HD / 16;     // n-tiles of a head's qkv slice / of one of q,k,v
    constexpr int WQ_PIECES = 3 * HD * PPR, NWQ = (WQ_PIECES + 255) / 256;
    constexpr int WP_PIECES = 96 * PPR, NWP = (WP_PIECES + 255) / 256;
    constexpr int NPC = C / 96;                          // proj chunks of 96 output features

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;          // [RP][LDX]   normalised x; later the output tile
    _Float16* Os = Xs + K::XS;               // [RP][LDX]   attention output, all heads
    _Float16* Qs = Os + K::OS;               // [RPQ][LDQ]
    _Float16* Ks = Qs + K::QS;               // [RPQ][LDQ]
    _Float16* VTs = Ks + K::QS;              // [HD][LDV]   v transposed: [feature][token]
    _Float16* Ws = VTs + K::VS;              // [WROWS][LDX]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const long win0 = (long)blockIdx.x * G;                 // first global window of this workgroup
    const long total_win = (long)p.B * p.nwin;
    const int HW = p.nwin * NTOK;
    const _Float16* __restrict__ X = (const _Float16*)p.x;
    const _Float16* __restrict__ Wqkv = (const _Float16*)p.wqkv;   // [3C][C]
    const _Float16* __restrict__ Wproj = (const _Float16*)p.wproj; // [C][C]

    // ---- weight prefetch helpers (global -> registers -> LDS)
    u32x4 rq[NWQ];
#define W2X_SA_PREFETCH_QKV(H)                                                                               \
    {                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < NWQ; ++t) {                                                    \
            const int idx = tid + t * 256;                                                                   \
            if (WQ_PIECES % 256 == 0 || idx < WQ_PIECES) {                                                   \
                const int rr = idx / PPR, kp = idx - rr * PPR;                                               \
                const int grow = (rr / HD) * C + (H) * HD + (rr % HD);                                       \
                rq[t] = *(const u32x4*)(Wqkv + (size_t)grow * C + kp * 8);                                   \
            }                                                                                                \
        }                                                                                                    \
    }
#define W2X_SA_STAGE_QKV()                                                                                   \
    {                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < NWQ; ++t) {                                                    \
            const int idx = tid + t * 256;                                                                   \
            if (WQ_PIECES % 256 == 0 || idx < WQ_PIECES) {                                                   \
                const int rr = idx / PPR, kp = idx - rr * PPR;                                               \
                *(u32x4*)(Ws + rr * LDX + kp * 8) = rq[t];                                                   \
            }                                                                                                \
        }                                                                                                    \
    }
    W2X_SA_PREFETCH_QKV(0);

    // ---- gather + LayerNorm into Xs; zero what the MFMAs may touch beyond the written rows
    int my_pix[NPASS];   // source pixel row of this thread's row in each pass (-1: none)
    {
        const int li = tid & (LPR - 1);
        half8 xr[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            int pix = -1;
            if (r < R) {
                const long wr = win0 * NTOK + r;             // global window-order row
                if (wr < total_win * NTOK) { const long b = wr / HW; const int ml = (int)(wr - b * HW); pix = (int)(b * HW) + p.table[ml]; }
            }
            my_pix[ps] = pix;
            half8 h = {};
            if (pix >= 0 && li < PPR) h = *(const half8*)(X + (size_t)pix * C + li * 8);
            xr[ps] = h;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)xr[ps][e];
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e];
            s = group_sum<LPR>(s);
            const float mean = s * (1.f / C);
            float q = 0.f;
            if (li < PPR) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] -= mean; q += v[e] * v[e]; }
            }
            q = group_sum<LPR>(q);
            const float rstd = rsqrtf(q * (1.f / C) + p.eps);
            if (r < RP && li < PPR) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (_Float16)(v[e] * rstd);   // rows without a source are exact zeros
                *(half8*)(Xs + r * LDX + li * 8) = o;
            }
        }
        // pad tokens of v^T must be finite (they are multiplied by P = 0): zero columns [RP, RPQ)
        for (int i = tid; i < HD * (RPQ - RP); i += 256) { const int d = i / (RPQ - RP), c = i - d * (RPQ - RP); VTs[d * LDV + RP + c] = (_Float16)0.f; }
    }

    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const half8 zero8 = {};

#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
        __syncthreads();                 // previous head's attention is done with Qs/Ks/VTs; Xs is complete
        W2X_SA_STAGE_QKV();
        __syncthreads();
        if (h + 1 < NH) W2X_SA_PREFETCH_QKV(h + 1);
        // ---- q,k,v of this head for all rows: q,k transposed (rows = features), v normal (rows = tokens)
        for (int mt = wv; mt < RT; mt += 4) {
            float4v acc[NTQ];
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) acc[nt] = zero4;
#pragma unroll
            for (int ks = 0; ks < C / 32; ++ks) {
                const half8 xf = *(const half8*)(Xs + (mt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
                for (int nt = 0; nt < NTQ; ++nt) {
                    const half8 wf = *(const half8*)(Ws + (nt * 16 + fr) * LDX + ks * 32 + g * 8);
                    if (nt < 2 * NTH) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[nt], 0, 0, 0);
                    else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, wf, acc[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) {
                const int which = nt / NTH, ntl = nt - which * NTH;      // 0 q, 1 k, 2 v
                if (which < 2) {
                    const float4v b = *(const float4v*)(p.bqkv + which * C + h * HD + ntl * 16 + g * 4);
                    const float sc = which == 0 ? p.scale : 1.f;
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)((acc[nt][j] + b[j]) * sc);
                    *(half4*)((which == 0 ? Qs : Ks) + (mt * 16 + fr) * LDQ + ntl * 16 + g * 4) = o;
                } else {
                    const float b = p.bqkv[2 * C + h * HD + ntl * 16 + fr];
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)(acc[nt][j] + b);
                    *(half4*)(VTs + (ntl * 16 + fr) * LDV + mt * 16 + g * 4) = o;
                }
            }
        }
        __syncthreads();
        // ---- attention of this head: C=96: wave = window, 3 query tiles; C=192: two waves per window (query tiles {0,1} / {2})
        {
            const int w = G == 4 ? wv : (wv >> 1);
            const int qt0 = G == 4 ? 0 : ((wv & 1) ? 2 : 0);
            const int qt1 = G == 4 ? 3 : ((wv & 1) ? 3 : 2);
            const long gw = win0 + w;
            if (gw < total_win) {
                const int wl = (int)(gw % p.nwin);
                const _Float16* bias = (const _Float16*)p.bias + ((size_t)p.maskid[wl] * NH + h) * NTOK * NTOK;
                const int rbase = w * NTOK;
                half8 kf[3];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) kf[kt] = (g * 8 < HD) ? *(const half8*)(Ks + (rbase + kt * 16 + fr) * LDQ + g * 8) : zero8;
                for (int qt = qt0; qt < qt1; ++qt) {
                    const half8 qf = (g * 8 < HD) ? *(const half8*)(Qs + (rbase + qt * 16 + fr) * LDQ + g * 8) : zero8;
                    const int query = qt * 16 + fr;
                    const int qrow = query < NTOK ? query : NTOK - 1;
                    float s[3][4];
                    float mx = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        const float4v a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt], qf, zero4, 0, 0, 0);   // rows = keys, cols = queries
                        const int k0 = kt * 16 + g * 4;
                        half4 bv = {};
                        if (k0 < NTOK) bv = *(const half4*)(bias + qrow * NTOK + k0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = (k0 + j < NTOK) ? a[j] + (float)bv[j] : -INFINITY;
                            s[kt][j] = v; mx = fmaxf(mx, v);
                        }
                    }
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float l = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float e = __expf(s[kt][j] - mx); s[kt][j] = e; l += e; }
                    l += __shfl_xor(l, 16);
                    l += __shfl_xor(l, 32);
                    const float inv = __builtin_amdgcn_rcpf(l);
                    const half8 pf0 = {(_Float16)s[0][0], (_Float16)s[0][1], (_Float16)s[0][2], (_Float16)s[0][3],
                                       (_Float16)s[1][0], (_Float16)s[1][1], (_Float16)s[1][2], (_Float16)s[1][3]};
                    const half8 pf1 = {(_Float16)s[2][0], (_Float16)s[2][1], (_Float16)s[2][2], (_Float16)s[2][3],
                                       (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
                    for (int dt = 0; dt < NTH; ++dt) {
                        const _Float16* vp = VTs + (dt * 16 + fr) * LDV + rbase + g * 4;
                        const half4 v0 = *(const half4*)vp, v1 = *(const half4*)(vp + 16), v2 = *(const half4*)(vp + 32);
                        const half8 vf0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        const half8 vf1 = {v2[0], v2[1], v2[2], v2[3], (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                        float4v o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf0, pf0, zero4, 0, 0, 0);   // rows = features, cols = queries
                        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf1, pf1, o, 0, 0, 0);
                        if (query < NTOK) {
                            half4 oh;
#pragma unroll
                            for (int j = 0; j < 4; ++j) oh[j] = (_Float16)(o[j] * inv);
                            *(half4*)(Os + (rbase + query) * LDX + h * HD + dt * 16 + g * 4) = oh;
                        }
                    }
                }
            }
        }
    }

    // ---- proj: out = Os * Wproj^T + b, 96 output features per staged chunk; the tile is written over Xs
    u32x4 rp[NWP];
#pragma unroll 1
    for (int pc = 0; pc < NPC; ++pc) {
#pragma unroll
        for (int t = 0; t < NWP; ++t) {
            const int idx = tid + t * 256;
            if (WP_PIECES % 256 == 0 || idx < WP_PIECES) { const int rr = idx / PPR, kp = idx - rr * PPR; rp[t] = *(const u32x4*)(Wproj + (size_t)(pc * 96 + rr) * C + kp * 8); }
        }
        __syncthreads();     // all waves done with Ws (last head's qkv / previous chunk) and, for pc = 0, with Os writes
#pragma unroll
        for (int t = 0; t < NWP; ++t) {
            const int idx = tid + t * 256;
            if (WP_PIECES % 256 == 0 || idx < WP_PIECES) { const int rr = idx / PPR, kp = idx - rr * PPR; *(u32x4*)(Ws + rr * LDX + kp * 8) = rp[t]; }
        }
        __syncthreads();
        for (int mt = wv; mt < RT; mt += 4) {
            float4v acc[6];
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) acc[nt] = zero4;
#pragma unroll
            for (int ks = 0; ks < C / 32; ++ks) {
                const half8 of = *(const half8*)(Os + (mt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) {
                    const half8 wf = *(const half8*)(Ws + (nt * 16 + fr) * LDX + ks * 32 + g * 8);
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(of, wf, acc[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) {
                const float b = p.bproj[pc * 96 + nt * 16 + fr];
#pragma unroll
                for (int j = 0; j < 4; ++j) Xs[(mt * 16 + g * 4 + j) * LDX + pc * 96 + nt * 16 + fr] = (_Float16)(acc[nt][j] + b);
            }
        }
    }
    __syncthreads();

    // ---- row pieces: + residual x (same gather table), scatter store, LayerNorm statistics for the next op
    {
        _Float16* __restrict__ Y = (_Float16*)p.y;
        const int li = tid & (LPR - 1);
        half8 xres[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            half8 h = {};
            if (my_pix[ps] >= 0 && li < PPR) h = *(const half8*)(X + (size_t)my_pix[ps] * C + li * 8);
            xres[ps] = h;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            const int pix = my_pix[ps];
            const bool ok = pix >= 0 && li < PPR;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (ok) {
                const half8 c = *(const half8*)(Xs + r * LDX + li * 8);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) { o[e] = (_Float16)((float)c[e] + (float)xres[ps][e]); v[e] = (float)o[e]; }
                *(half8*)(Y + (size_t)pix * C + li * 8) = o;
            }
            if (p.stats_out) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
                s = group_sum<LPR>(s);
                const float mean = s * (1.f / C);
                float q = 0.f;
                if (ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
                }
                q = group_sum<LPR>(q);
                if (ok && li == 0) { p.stats_out[2 * (size_t)pix] = mean; p.stats_out[2 * (size_t)pix + 1] = rsqrtf(q * (1.f / C) + p.eps_out); }
            }
        }
    }
#undef W2X_SA_PREFETCH_QKV
#undef W2X_SA_STAGE_QKV
}

template <int C, int HD>
hipError_t launch_sa(const SwinAttnParams& p, hipStream_t s) {
    using K = SwinCfg<C, HD>;
    auto kern = swin_attn_kernel<C, HD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K::SMEM);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const long total_win = (long)p.B * p.nwin;
    dim3 grid((unsigned)((total_win + K::G - 1) / K::G));
    hipLaunchKernelGGL(kern, grid, dim3(256), K::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

bool swin_attn_supported(int C, int heads, int hd, int ws) {
    return ws == 6 && heads * hd == C && ((C == 96 && hd == 16) || (C == 192 && hd == 32));
}

hipError_t launch_swin_attn(const SwinAttnParams& p, hipStream_t s) {
    if (p.C == 96 && p.hd == 16) return launch_sa<96, 16>(p, s);
    if (p.C == 192 && p.hd == 32) return launch_sa<192, 32>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace w2x
