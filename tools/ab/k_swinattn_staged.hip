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
#include <cstdlib>

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// packed-fp16 row helpers: sum and sum of squares of 8 halves through v_dot2_f32_f16 (fp32 accumulation), and the
// LayerNorm affine (x - mean) * rstd as one mixed-precision fma per element (fp16 in, fp32 scale/offset, one rounding)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sum_sq8(const half8 v, float& s, float& q) {
    const half2v one = {(_Float16)1.f, (_Float16)1.f};
    s = 0.f; q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const half2v h = {v[2 * k], v[2 * k + 1]};
        s = __builtin_amdgcn_fdot2(h, one, s, false);
        q = __builtin_amdgcn_fdot2(h, h, q, false);
    }
}
__device__ __forceinline__ half8 norm8(const half8 v, float rstd, float nm) {
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)fmaf((float)v[e], rstd, nm);
    return o;
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    if (LPR == 32) v += __shfl_xor(v, 16);
    return v;
}

// max / sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with v_permlane16/32_swap (no LDS traffic).
// The swap exchanges rows/halves between TWO registers holding the same value; written as inline asm because the
// __builtin_amdgcn_permlane*_swap pair-return folds both results onto one register when the inputs are equal
// (checked on hardware).  "s_nop 1" = the two wait states a VALU write needs before a permlane reads it.
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_max(float v) {
    float a = v, b = v; swap16(a, b); v = fmaxf(a, b);
    a = v; b = v; swap32(a, b); return fmaxf(a, b);
}
__device__ __forceinline__ float rows_sum(float v) {
    float a = v, b = v; swap16(a, b); v = a + b;
    a = v; b = v; swap32(a, b); return a + b;
}

// Diagnostic build only (W2X_STAMPS=1 at run time selects it): per-phase s_memtime deltas summed over all waves.
// g_sa_stamps[k]: 0 gather+LN, 1 barrier A + stage + barrier B, 2 q/k/v products, 3 barrier C, 4 attention, 5 proj (incl. barriers),
// 6 final row pass, 7 waves counted.
__device__ unsigned long long g_sa_stamps[8];
#define W2X_STAMP(K)                                                                                        \
    if (STAMPS) {                                                                                           \
        unsigned long long t_;                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (lane == 0) atomicAdd(&g_sa_stamps[K], t_ - tprev);                                              \
        tprev = t_;                                                                                         \
    }

template <int C, int HD>
struct SwinCfg {
    static constexpr int NTOK = 36;
    static constexpr int G = 2;                          // windows per workgroup
    static constexpr int HPI = HD == 16 ? 2 : 1;         // heads per iteration: 3*HD*HPI = 96 weight rows per staged slice
    static constexpr int NW = C == 96 ? 4 : 8;           // waves per workgroup: C=96 runs 2 workgroups per CU, C=192 (124 KB LDS) one
    static constexpr int NT = NW * 64;
    static constexpr int R = G * NTOK;                   // token rows (72)
    static constexpr int RT = (R + 15) / 16, RP = RT * 16;   // 5 tiles, 80 rows
    static constexpr int RPQ = RP + 16;                  // rows incl. the overrun of the last window's third 16-row tile
    static constexpr int NH = C / HD;
    static constexpr int LDX = C + 8, LDQ = HPI * HD + 8, LDV = RPQ + 8;
    static constexpr int WROWS = 96;
    static constexpr int XS = RP * LDX, OS = RP * LDX, QS = RPQ * LDQ, VS = HPI * HD * LDV, WS = WROWS * LDX;
    static constexpr int SMEM = (XS + OS + 2 * QS + VS + WS) * 2 + 96 * 4;   // + the staged slice of the qkv bias
};

template <int C, int HD, bool STAMPS>
__global__ __launch_bounds__((C == 96 ? 256 : 512), 2) void swin_attn_kernel(const SwinAttnParams p) {
    using K = SwinCfg<C, HD>;
    constexpr int NTOK = K::NTOK, G = K::G, R = K::R, RT = K::RT, RP = K::RP, RPQ = K::RPQ, NH = K::NH, HPI = K::HPI;
    static_assert(3 * HD * HPI == 96 && G == 2, "unit mapping below assumes 6 n-tiles per slice and 2 windows");
    constexpr int LDX = K::LDX, LDQ = K::LDQ, LDV = K::LDV;
    constexpr int NW = K::NW, NT = K::NT;
    constexpr int LPR = C == 96 ? 16 : 32, PPR = C / 8, RPP = NT / LPR;
    constexpr int NPASS = (RP + RPP - 1) / RPP;
    constexpr int NTH = HD / 16;                         // 16-wide tiles per head of one of q,k,v
    constexpr int WQ_PIECES = 96 * PPR, NWQ = (WQ_PIECES + NT - 1) / NT;
    constexpr int WP_PIECES = 96 * PPR, NWP = (WP_PIECES + NT - 1) / NT;
    constexpr int NPC = C / 96;                          // proj chunks of 96 output features

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;          // [RP][LDX]   normalised x; later the output tile
    _Float16* Os = Xs + K::XS;               // [RP][LDX]   attention output, all heads
    _Float16* Qs = Os + K::OS;               // [RPQ][LDQ]
    _Float16* Ks = Qs + K::QS;               // [RPQ][LDQ]
    _Float16* VTs = Ks + K::QS;              // [HPI*HD][LDV]   v transposed: [feature][token]
    _Float16* Ws = VTs + K::VS;              // [WROWS][LDX]
    float* Bs = (float*)(Ws + K::WS);        // [96] qkv bias of the staged slice, same row order as Ws

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): unit loops branch on scalars
    const int fr = lane & 15, g = lane >> 4;
    const long win0 = (long)blockIdx.x * G;                 // first global window of this workgroup
    const long total_win = (long)p.B * p.nwin;
    const int HW = p.nwin * NTOK;
    const _Float16* __restrict__ X = (const _Float16*)p.x;
    const _Float16* __restrict__ Wqkv = (const _Float16*)p.wqkv;   // [3C][C]
    const _Float16* __restrict__ Wproj = (const _Float16*)p.wproj; // [C][C]

    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const half8 zero8 = {};
    unsigned long long tprev = 0;
    if (STAMPS) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); if (lane == 0) atomicAdd(&g_sa_stamps[7], 1ull); }

    // ---- weight prefetch helpers (global -> registers -> LDS)
    u32x4 rq[NWQ];
    u32x4 rp[NWP];
    float rbias = 0.f;
#define W2X_SA_PREFETCH_QKV(H)                                                                               \
    {                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < NWQ; ++t) {                                                    \
            const int idx = tid + t * NT;                                                                    \
            if (WQ_PIECES % NT == 0 || idx < WQ_PIECES) {                                                    \
                const int rr = idx / PPR, kp = idx - rr * PPR;                                               \
                const int grow = (rr / (HPI * HD)) * C + (H) * HD + (rr % (HPI * HD));                       \
                rq[t] = *(const u32x4*)(Wqkv + (size_t)grow * C + kp * 8);                                   \
            }                                                                                                \
        }                                                                                                    \
        if (tid < 96) rbias = p.bqkv[(tid / (HPI * HD)) * C + (H) * HD + (tid % (HPI * HD))];                \
    }
#define W2X_SA_STAGE_QKV()                                                                                   \
    {                                                                                                        \
        _Pragma("unroll") for (int t = 0; t < NWQ; ++t) {                                                    \
            const int idx = tid + t * NT;                                                                    \
            if (WQ_PIECES % NT == 0 || idx < WQ_PIECES) {                                                    \
                const int rr = idx / PPR, kp = idx - rr * PPR;                                               \
                *(u32x4*)(Ws + rr * LDX + kp * 8) = rq[t];                                                   \
            }                                                                                                \
        }                                                                                                    \
        if (tid < 96) Bs[tid] = rbias;                                                                       \
    }
    W2X_SA_PREFETCH_QKV(0);

    // the two windows of this workgroup: batch item and table offset (wave-uniform 32-bit arithmetic)
    const int iw0 = (int)win0, iw1 = iw0 + 1, itotal = (int)total_win;
    const bool wok0 = iw0 < itotal, wok1 = iw1 < itotal;
    const int wb0 = iw0 / p.nwin, wb1 = iw1 / p.nwin;
    const int pixbase0 = wb0 * HW, pixbase1 = wb1 * HW;
    const int wl0 = iw0 - wb0 * p.nwin, wl1 = iw1 - wb1 * p.nwin;      // window index inside its batch item
    const int tabbase0 = wl0 * NTOK, tabbase1 = wl1 * NTOK;
    const int nwx = p.W / 6;
    const int wy0 = wl0 / nwx, wx0 = wl0 - wy0 * nwx, wy1 = wl1 / nwx, wx1 = wl1 - wy1 * nwx;

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
                const int w = r >= NTOK ? 1 : 0;             // G == 2 windows per workgroup
                if (w == 0 ? wok0 : wok1) {
                    const int t = r - w * NTOK;
                    if (p.ry >= 0) {   // closed form of the window table: no dependent lookup in front of the x loads
                        const int ty = t / 6, tx = t - ty * 6;
                        int y = (w == 0 ? wy0 : wy1) * 6 + ty + p.ry, x = (w == 0 ? wx0 : wx1) * 6 + tx + p.rx;
                        y -= y >= p.H ? p.H : 0; x -= x >= p.W ? p.W : 0;
                        pix = (w == 0 ? pixbase0 : pixbase1) + y * p.W + x;
                    } else pix = (w == 0 ? pixbase0 : pixbase1) + p.table[(w == 0 ? tabbase0 : tabbase1) + t];
                }
            }
            my_pix[ps] = pix;
            half8 h = {};
            if (pix >= 0 && li < PPR) h = *(const half8*)(X + (size_t)pix * C + li * 8);
            xr[ps] = h;
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            float s, q;
            sum_sq8(xr[ps], s, q);                       // lanes beyond the row and rows without a source hold zeros
            s = group_sum<LPR>(s);
            q = group_sum<LPR>(q);
            const float mean = s * (1.f / C);
            const float rstd = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps);
            if (r < RP && li < PPR) *(half8*)(Xs + r * LDX + li * 8) = norm8(xr[ps], rstd, -mean * rstd);   // source-less rows: 0*rstd - 0 = 0
        }
        // rows [RP, RPQ) of q/k are only ever read as masked keys; they must be finite (bias = -inf is added to them)
        for (int i = tid; i < (RPQ - RP) * LDQ; i += NT) { Qs[RP * LDQ + i] = (_Float16)0.f; Ks[RP * LDQ + i] = (_Float16)0.f; }
        // pad tokens of v^T must be finite (they are multiplied by P = 0): zero columns [RP, RPQ)
        for (int i = tid; i < HPI * HD * (RPQ - RP); i += NT) { const int d = i / (RPQ - RP), c = i - d * (RPQ - RP); VTs[d * LDV + RP + c] = (_Float16)0.f; }
    }


    // attention unit of this wave
    //   NW == 4 (HPI == 2): wave = (window, head of the pair), all three query tiles
    //   NW == 8 (HPI == 1): waves 0..5 = (window, query tile), waves 6,7 idle in this phase
    const int aw = NW == 4 ? (wv >> 1) : (wv < 6 ? wv / 3 : 0);
    const int ahp = NW == 4 ? (wv & 1) : 0;
    const int qt0 = NW == 4 ? 0 : wv % 3;
    const int qt1 = NW == 4 ? 3 : qt0 + 1;
    constexpr int NQI = NW == 4 ? 3 : 1;                  // query tiles per unit
    const long agw = win0 + aw;
    const bool aok = agw < total_win && (NW == 4 || wv < 6);
    const int amask = aok ? p.maskid[(int)(agw % p.nwin)] : 0;

#pragma unroll
    for (int h = 0; h < NH; h += HPI) {   // fully unrolled: the weight prefetch registers are renamed statically, so the loads issued
                                          // for iteration h+1 are only waited for when that iteration stages them
        // relative-position bias (+ shift mask) of this wave's unit: fp32, pre-multiplied by log2(e), stored in load order.
        // Fetched now, used as the initial accumulator of the S^T products after the q,k,v phase.
        float4v bv[NQI][3];
        {
            const float* bias = p.bias32 + ((size_t)amask * NH + h + ahp) * (3 * 576) + lane * 4;   // load-order layout, see lower.cpp
#pragma unroll
            for (int qi = 0; qi < NQI; ++qi)
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const int qt = qt0 + qi;
                    float4v t = zero4;
                    if (qt < qt1) { if (kt < 2) t = *(const float4v*)(bias + qt * 576 + kt * 256); else t[0] = bias[qt * 576 + 512 - lane * 3]; }
                    bv[qi][kt] = t;
                }
        }
        W2X_STAMP(h == 0 ? 0 : 4)
        __syncthreads();                 // previous iteration's attention is done with Qs/Ks/VTs; Xs is complete
        W2X_SA_STAGE_QKV();
        __syncthreads();
        W2X_STAMP(1)
        if (h + HPI < NH) W2X_SA_PREFETCH_QKV(h + HPI)
        else {   // last iteration: fetch the first proj weight chunk now, it is staged after the attention phase
#pragma unroll
            for (int t = 0; t < NWP; ++t) {
                const int idx = tid + t * NT;
                if (WP_PIECES % NT == 0 || idx < WP_PIECES) { const int rr = idx / PPR, kp = idx - rr * PPR; rp[t] = *(const u32x4*)(Wproj + (size_t)rr * C + kp * 8); }
            }
        }
        // ---- q,k,v of these heads for all rows.  15 units of (16-row tile, pair of 16-wide n-tiles): pair 0 = q, 1 = k
        //      (computed transposed: rows = features), 2 = v (rows = tokens)
        for (int u = wv; u < RT * 3; u += NW) {
            const int mt = u / 3, np = u - mt * 3;
            float4v acc[2] = {zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < C / 32; ++ks) {
                const half8 xf = *(const half8*)(Xs + (mt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const half8 wf = *(const half8*)(Ws + ((np * 2 + t) * 16 + fr) * LDX + ks * 32 + g * 8);
                    if (np < 2) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, wf, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int f0 = t * 16;                               // feature offset inside the slice's q / k / v block (HPI*HD = 32 wide)
                if (np < 2) {
                    const float4v b = *(const float4v*)(Bs + np * (HPI * HD) + f0 + g * 4);
                    const float sc = np == 0 ? p.scale * 1.44269504088896341f : 1.f;   // log2(e) folded into q: softmax uses exp2
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)((acc[t][j] + b[j]) * sc);
                    *(half4*)((np == 0 ? Qs : Ks) + (mt * 16 + fr) * LDQ + f0 + g * 4) = o;
                } else {
                    const float b = Bs[2 * (HPI * HD) + f0 + fr];
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (_Float16)(acc[t][j] + b);
                    *(half4*)(VTs + (f0 + fr) * LDV + mt * 16 + g * 4) = o;
                }
            }
        }
        W2X_STAMP(2)
        __syncthreads();
        W2X_STAMP(3)
        // ---- attention of this wave's unit: the query tiles are processed in phases so their chains interleave
        if (aok) {
            const int rbase = aw * NTOK;
            const int fo = ahp * HD;                                  // feature offset of this unit's head inside Qs/Ks/VTs
            half8 kf[3];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) kf[kt] = (g * 8 < HD) ? *(const half8*)(Ks + (rbase + kt * 16 + fr) * LDQ + fo + g * 8) : zero8;
            // third key tile: only keys 32..35 exist.  They are placed on tile rows 0,4,8,12 (one per lane group g, register
            // j = 0), so the softmax below touches 9 instead of 12 values per lane; rows 4g+1..3 repeat key 32+g and are ignored.
            kf[2] = (g * 8 < HD) ? *(const half8*)(Ks + (rbase + 32 + (fr >> 2)) * LDQ + fo + g * 8) : zero8;
            float4v s[NQI][3];
#pragma unroll
            for (int qi = 0; qi < NQI; ++qi) {
                if (qt0 + qi < qt1) {
                    const half8 qf = (g * 8 < HD) ? *(const half8*)(Qs + (rbase + (qt0 + qi) * 16 + fr) * LDQ + fo + g * 8) : zero8;
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt)   // rows = keys, cols = queries; accumulator starts at the bias (shift mask included)
                        s[qi][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt], qf, bv[qi][kt], 0, 0, 0);
                }
            }
            float inv[NQI];
            half8 pf0[NQI], pf1[NQI];
#pragma unroll
            for (int qi = 0; qi < NQI; ++qi) {
                if (qt0 + qi < qt1) {
                    float mx = fmaxf(fmaxf(s[qi][0][0], s[qi][0][1]), fmaxf(s[qi][0][2], s[qi][0][3]));
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[qi][1][j]);
                    mx = rows_max(fmaxf(mx, s[qi][2][0]));
                    float l = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float e = __builtin_amdgcn_exp2f(s[qi][kt][j] - mx); s[qi][kt][j] = e; l += e; }
                    { const float e = __builtin_amdgcn_exp2f(s[qi][2][0] - mx); s[qi][2][0] = e; l += e; }
                    l = rows_sum(l);
                    inv[qi] = __builtin_amdgcn_rcpf(l);
                    pf0[qi] = (half8){(_Float16)s[qi][0][0], (_Float16)s[qi][0][1], (_Float16)s[qi][0][2], (_Float16)s[qi][0][3],
                                      (_Float16)s[qi][1][0], (_Float16)s[qi][1][1], (_Float16)s[qi][1][2], (_Float16)s[qi][1][3]};
                    pf1[qi] = (half8){(_Float16)s[qi][2][0], (_Float16)0.f, (_Float16)0.f, (_Float16)0.f,
                                      (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                }
            }
#pragma unroll
            for (int dt = 0; dt < NTH; ++dt) {
                const _Float16* vp = VTs + (fo + dt * 16 + fr) * LDV + rbase + g * 4;
                const half4 v0 = *(const half4*)vp, v1 = *(const half4*)(vp + 16);
                const _Float16 v2 = VTs[(fo + dt * 16 + fr) * LDV + rbase + 32 + g];      // key 32+g: k slot 8g of the second product
                const half8 vf0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const half8 vf1 = {v2, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
                for (int qi = 0; qi < NQI; ++qi) {
                    if (qt0 + qi < qt1) {
                        float4v o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf0, pf0[qi], zero4, 0, 0, 0);   // rows = features, cols = queries
                        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf1, pf1[qi], o, 0, 0, 0);
                        const int query = (qt0 + qi) * 16 + fr;
                        if (query < NTOK) {
                            half4 oh;
#pragma unroll
                            for (int j = 0; j < 4; ++j) oh[j] = (_Float16)(o[j] * inv[qi]);
                            *(half4*)(Os + (rbase + query) * LDX + (h + ahp) * HD + dt * 16 + g * 4) = oh;
                        }
                    }
                }
            }
        }
    }

    W2X_STAMP(4)
    // ---- proj: out = Os * Wproj^T + b, 96 output features per staged chunk; the tile is written over Xs
#pragma unroll 1
    for (int pc = 0; pc < NPC; ++pc) {
        if (pc > 0) {
#pragma unroll
            for (int t = 0; t < NWP; ++t) {
                const int idx = tid + t * NT;
                if (WP_PIECES % NT == 0 || idx < WP_PIECES) { const int rr = idx / PPR, kp = idx - rr * PPR; rp[t] = *(const u32x4*)(Wproj + (size_t)(pc * 96 + rr) * C + kp * 8); }
            }
        }
        __syncthreads();     // all waves done with Ws (last head's qkv / previous chunk) and, for pc = 0, with Os writes
#pragma unroll
        for (int t = 0; t < NWP; ++t) {
            const int idx = tid + t * NT;
            if (WP_PIECES % NT == 0 || idx < WP_PIECES) { const int rr = idx / PPR, kp = idx - rr * PPR; *(u32x4*)(Ws + rr * LDX + kp * 8) = rp[t]; }
        }
        __syncthreads();
        for (int u = wv; u < RT * 2; u += NW) {
            const int mt = u >> 1, n3 = (u & 1) * 3;
            float4v acc[3] = {zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < C / 32; ++ks) {
                const half8 of = *(const half8*)(Os + (mt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const half8 wf = *(const half8*)(Ws + ((n3 + t) * 16 + fr) * LDX + ks * 32 + g * 8);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(of, wf, acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const float b = p.bproj[pc * 96 + (n3 + t) * 16 + fr];
#pragma unroll
                for (int j = 0; j < 4; ++j) Xs[(mt * 16 + g * 4 + j) * LDX + pc * 96 + (n3 + t) * 16 + fr] = (_Float16)(acc[t][j] + b);
            }
        }
    }
    __syncthreads();
    W2X_STAMP(5)

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
            half8 o = {};
            if (ok) {
                o = *(const half8*)(Xs + r * LDX + li * 8) + xres[ps];     // fp16 + fp16 rounded once == fp32 add rounded to fp16
                *(half8*)(Y + (size_t)pix * C + li * 8) = o;
            }
            if (p.stats_out) {
                float s, q;
                sum_sq8(o, s, q);
                s = group_sum<LPR>(s);
                q = group_sum<LPR>(q);
                const float mean = s * (1.f / C);
                if (ok && li == 0) { p.stats_out[2 * (size_t)pix] = mean; p.stats_out[2 * (size_t)pix + 1] = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out); }
            }
        }
    }
    W2X_STAMP(6)
#undef W2X_SA_PREFETCH_QKV
#undef W2X_SA_STAGE_QKV
}

template <int C, int HD>
hipError_t launch_sa(const SwinAttnParams& p, hipStream_t s) {
    using K = SwinCfg<C, HD>;
    static const bool stamps = getenv("W2X_STAMPS") != nullptr;
    auto kern = stamps ? swin_attn_kernel<C, HD, true> : swin_attn_kernel<C, HD, false>;
    static unsigned lds_ok = 0, lds_ok2 = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)swin_attn_kernel<C, HD, true>, K::SMEM, lds_ok); e != hipSuccess) return e;
    if (hipError_t e = ensure_dynamic_lds((const void*)swin_attn_kernel<C, HD, false>, K::SMEM, lds_ok2); e != hipSuccess) return e;
    const long total_win = (long)p.B * p.nwin;
    dim3 grid((unsigned)((total_win + K::G - 1) / K::G));
    hipLaunchKernelGGL(kern, grid, dim3(K::NT), K::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t read_swin_attn192_stamps(unsigned long long* out);   // k_swinattn192.hip

// copies and clears the diagnostic stamp counters: out[0..7] this file's kernel, out[8..15] the C = 192 register-resident one
hipError_t read_swin_attn_stamps(unsigned long long* out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sa_stamps), sizeof(unsigned long long) * 8);
    if (e != hipSuccess) return e;
    unsigned long long z[8] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_sa_stamps), z, sizeof(z));
    if (e != hipSuccess) return e;
    return read_swin_attn192_stamps(out + 8);
}

bool swin_attn_supported(int C, int heads, int hd, int ws) {
    return ws == 6 && heads * hd == C && ((C == 96 && hd == 16) || (C == 192 && hd == 32));
}

hipError_t launch_swin_attn96(const SwinAttnParams& p, hipStream_t s);    // k_swinattn96.hip
hipError_t launch_swin_attn192(const SwinAttnParams& p, hipStream_t s);   // k_swinattn192.hip

hipError_t launch_swin_attn(const SwinAttnParams& p, hipStream_t s) {
    static const bool v1 = getenv("W2X_SA_V1") != nullptr;   // A/B switch: the barrier-staged kernel of this file
    const bool staged = v1 || !p.wqkv_frag || !p.wproj_frag;
    if (p.C == 96 && p.hd == 16) return staged ? launch_sa<96, 16>(p, s) : launch_swin_attn96(p, s);
    if (p.C == 192 && p.hd == 32) return staged ? launch_sa<192, 32>(p, s) : launch_swin_attn192(p, s);
    return hipErrorInvalidValue;
}

}  // namespace w2x
