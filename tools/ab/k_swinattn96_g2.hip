// Round-1 schedule of the C = 96 fused attention (a workgroup = 2 windows, a wave = 3 heads of one window), kept for the A/B harness
// tools/ab/attn_ab.hip; the shipped kernel is waifu2x-tensorrt_amd/csrc/k_swinattn96.hip.
// Fused Swin attention branch, C = 96 / 6 heads of 16 / window 6x6, register-resident variant for gfx950.
//     y = x + proj( W-MSA( LayerNorm(x) ) )
// Same math, parameters and bias-table layout as k_swinattn.hip; the schedule is the one of k_swinattn192.hip:
// a workgroup = 4 waves = 2 windows; wave (w, hp) owns window w and the heads 2*it + hp (it = 0..2), reads its weights
// straight from L2 as MFMA fragments (fragment-major copy, one head ahead) and keeps q, k, v, S and P in registers.
// With a head dimension of 16 the accumulator layout (lane (col, g) holds rows 4g..4g+3) is exactly the operand layout
// of v_mfma_f32_16x16x16_f16 (lane (row|col, g) holds k = 4g..4g+3), so
//     q^T, k^T (rows = features)  -> B / A operands of S^T = K Q^T,
//     v (rows = tokens)           -> A operand of O^T = V^T P^T,
//     S^T after the softmax       -> B operand of O^T
// need no LDS round trip and no permutation.  The q/k/v products use v_mfma_f32_16x16x32_f16 over the 96 channels.
// LDS holds the normalised x slabs (48 rows per window: tokens 0..31, then tokens 32..35 on rows 32, 36, 40, 44 so that
// key 32+g sits on row 4g of the third key tile - see k_swinattn192.hip) and the head outputs for proj.
// Four workgroup barriers in the whole kernel.
// The kernel is VALU-issue bound (SQ counters: 11 VALU instructions per MFMA before, MFMA pipe < 25 % busy), so the vector
// work around the products is kept minimal: softmax denominators from a ones-operand MFMA, k bias dropped / v bias after the
// normalisation, packed fp32 (v_pk_*) where two values share an operation, interleaved permlane / DPP chains without wait
// states, the row -> pixel map computed once per workgroup, all row loads unconditional from clamped addresses.
#include "kernels.h"

namespace w2x {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

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
// sum over aligned groups of 16 lanes with DPP
__device__ __forceinline__ float group_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}
// Row-group sums of several independent values at once: v_add_f32 with a DPP operand (the compiler emits v_mov_dpp +
// v_add for v += dpp(v)); the chains are interleaved so that each one's two wait states between a VALU write and a
// DPP read are filled by the others.
#define W2X_DPP1(R, CTRL) "v_add_f32_dpp " R ", " R ", " R " " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define W2X_DPP4(CTRL) W2X_DPP1("%0", CTRL) W2X_DPP1("%1", CTRL) W2X_DPP1("%2", CTRL) W2X_DPP1("%3", CTRL)
#define W2X_DPP6(CTRL) W2X_DPP4(CTRL) W2X_DPP1("%4", CTRL) W2X_DPP1("%5", CTRL)
#define W2X_DPP_STEPS(N) "s_nop 1\n\t" W2X_DPP##N("quad_perm:[1,0,3,2]") W2X_DPP##N("quad_perm:[2,3,0,1]") W2X_DPP##N("row_half_mirror") W2X_DPP##N("row_mirror")
__device__ __forceinline__ void group_sum16_x4(float& a, float& b, float& c, float& d) {
    asm volatile(W2X_DPP_STEPS(4) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void group_sum16_x6(float& a, float& b, float& c, float& d, float& e, float& f) {
    asm volatile(W2X_DPP_STEPS(6) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
// Row maximum of three independent values at once: each chain's permlane wait states (two after the VALU write a swap
// reads, one before a VALU reads a swap's result) are filled by the other two chains, so the sequence carries no s_nop,
// and v_max_f32 is used as is (fmaxf() would canonicalise both swap results first).
__device__ __forceinline__ void rows_max3(float& a0, float& a1, float& a2) {
    float b0, b1, b2;
    asm volatile(
        "v_mov_b32 %3, %0\n\tv_mov_b32 %4, %1\n\tv_mov_b32 %5, %2\n\t"
        "v_permlane16_swap_b32 %0, %3\n\tv_permlane16_swap_b32 %1, %4\n\tv_permlane16_swap_b32 %2, %5\n\t"
        "v_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5\n\t"
        "v_mov_b32 %3, %0\n\tv_mov_b32 %4, %1\n\tv_mov_b32 %5, %2\n\t"
        "v_permlane32_swap_b32 %0, %3\n\tv_permlane32_swap_b32 %1, %4\n\tv_permlane32_swap_b32 %2, %5\n\t"
        "v_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5"
        : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(b0), "=&v"(b1), "=&v"(b2));
}

constexpr int C = 96, HD = 16, NH = 6, NTOK = 36, G = 2, R = G * NTOK, RT = 5, RP = RT * 16;
constexpr int SLAB = 48, RPX = G * SLAB;   // slab rows per window / in the tile
constexpr int LDX = C + 8;                 // 104 halves
constexpr int XS = RPX * LDX, OS = RP * LDX;
constexpr int PIXN = 80;                   // row -> pixel table entries (>= the rows the passes touch)
constexpr int SMEM96 = (XS + OS) * 2 + PIXN * 4;
constexpr int LPR = 16, PPR = C / 8, RPP = 256 / LPR, NPASS = (R + RPP - 1) / RPP;   // row passes: 16 lanes per row, 16 rows per pass, 5 passes

__device__ __forceinline__ int slab_row(int t) { return t < 32 ? t : 32 + 4 * (t - 32); }

__global__ __launch_bounds__(256, 3) void swin_attn96_g2_kernel(const SwinAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* Xs = (_Float16*)smem;              // [RPX][LDX] normalised x slabs; later the output tile [RP][LDX]
    _Float16* Os = Xs + XS;                      // [RP][LDX]  attention output, all heads, token order
    int* Pix = (int*)(Os + OS);                  // [PIXN] source pixel of each token row (-1: none); valid to the end

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    const int iw0 = blockIdx.x * G, iw1 = iw0 + 1, itotal = p.B * p.nwin;
    const bool wok0 = iw0 < itotal, wok1 = iw1 < itotal;
    const int HW = p.nwin * NTOK;
    const int wb0 = iw0 / p.nwin, wb1 = iw1 / p.nwin;
    const int pixbase0 = wb0 * HW, pixbase1 = wb1 * HW;
    const int wl0 = iw0 - wb0 * p.nwin, wl1 = iw1 - wb1 * p.nwin;
    const int nwx = p.W / 6;
    const int wy0 = wl0 / nwx, wx0 = wl0 - wy0 * nwx, wy1 = wl1 / nwx, wx1 = wl1 - wy1 * nwx;
    const _Float16* __restrict__ X = (const _Float16*)p.x;
    const _Float16* __restrict__ Wqkv = (const _Float16*)p.wqkv_frag;    // [18 row tiles][3 k-steps][64 lanes][8] (engine.cpp frag_major)
    const _Float16* __restrict__ Wproj = (const _Float16*)p.wproj_frag;  // [6 row tiles][3 k-steps][64 lanes][8]
    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const half8 zero8 = {};

    // this wave's attention unit
    const int aw = wv >> 1, ahp = wv & 1;
    const bool aok = aw == 0 ? wok0 : wok1;
    const int amask = aok ? p.maskid[aw == 0 ? wl0 : wl1] : 0;
    const int sbase = aw * SLAB, tbase = aw * NTOK;

    // weight fragments of head h (q, k, v rows h*16 + fr; 3 k-steps): lane holds [row][ks*32 + 8g .. +7]; stored
    // fragment-major, so each load is one contiguous KiB (row-major fragments touch 16 half cache lines each)
    half8 wr[2][9];
    const _Float16* wlane = Wqkv + lane * 8;
#define W2X_LOAD_W(SET, H)                                                                                   \
    {                                                                                                        \
        _Pragma("unroll") for (int m = 0; m < 3; ++m)                                                        \
            _Pragma("unroll") for (int ks = 0; ks < 3; ++ks)                                                 \
                wr[SET][m * 3 + ks] = *(const half8*)(wlane + (size_t)((m * NH + (H)) * 3 + ks) * 512);      \
    }
    W2X_LOAD_W(0, ahp);

    // ---- source pixel of every token row, worked out once per workgroup (one thread per row) and handed round in LDS
    {
        int pix = -1;
        if (tid < R) {
            const int w = tid >= NTOK ? 1 : 0;
            if (w == 0 ? wok0 : wok1) {
                const int t = tid - w * NTOK;
                if (p.ry >= 0) {
                    const int ty = t / 6, tx = t - ty * 6;
                    int y = (w == 0 ? wy0 : wy1) * 6 + ty + p.ry, x = (w == 0 ? wx0 : wx1) * 6 + tx + p.rx;
                    y -= y >= p.H ? p.H : 0; x -= x >= p.W ? p.W : 0;
                    pix = (w == 0 ? pixbase0 : pixbase1) + y * p.W + x;
                } else pix = (w == 0 ? pixbase0 : pixbase1) + p.table[(w == 0 ? wl0 : wl1) * NTOK + t];
            }
        }
        if (tid < PIXN) Pix[tid] = pix;
    }
    __syncthreads();

    // ---- gather + LayerNorm into the slabs
    {
        const int li = tid & (LPR - 1);
        half8 xr[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;          // token row 0..71 (the last pass is half empty)
            const int pix = Pix[r];
            // unconditional load from a clamped address (all passes in flight at once), zeroed afterwards
            half8 h = *(const half8*)(X + (size_t)(pix < 0 ? 0 : pix) * C + (li < PPR ? li : 0) * 8);
            if (!(pix >= 0 && li < PPR)) h = zero8;
            xr[ps] = h;
        }
        static_assert(NPASS == 5, "the row sums are reduced as 3 + 2 passes");
        float sm[NPASS], sq[NPASS];
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) sum_sq8(xr[ps], sm[ps], sq[ps]);
        group_sum16_x6(sm[0], sq[0], sm[1], sq[1], sm[2], sq[2]);
#pragma unroll
        for (int ps = 3; ps < 5; ++ps) sum_sq8(xr[ps], sm[ps], sq[ps]);
        group_sum16_x4(sm[3], sq[3], sm[4], sq[4]);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            const int w = r >= NTOK ? 1 : 0;
            const float mean = sm[ps] * (1.f / C);
            const float rstd = rsqrtf(fmaxf(sq[ps] * (1.f / C) - mean * mean, 0.f) + p.eps);
            if (r < R && li < PPR) *(half8*)(Xs + (w * SLAB + slab_row(r - w * NTOK)) * LDX + li * 8) = norm8(xr[ps], rstd, -mean * rstd);
        }
        // the 12 rows between tokens 32..35 of each slab are multiplied like the rest (results ignored): keep them finite
        for (int i = tid; i < G * 12 * PPR; i += 256) {
            const int rr = i / PPR, c = i - rr * PPR, w = rr / 12, k = rr - w * 12;
            *(half8*)(Xs + (w * SLAB + 33 + (k / 3) * 4 + (k % 3)) * LDX + c * 8) = zero8;
        }
    }
    __syncthreads();

    const float qscale = p.scale * 1.44269504088896341f;   // log2(e) folded into q: softmax uses exp2
    const int lane2 = g * 16 + (fr >> 2);                  // bias-table lane of the query this lane holds in query tile 2
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int h = 2 * it + ahp;
        const int cur = it & 1;
        if (it < 2) W2X_LOAD_W(cur ^ 1, h + 2);
        // bias (+ shift mask) of this unit in load order (lower.cpp): per query tile 2 x float4 + 1 float per lane;
        // it is the initial accumulator of S^T
        float4v s[3][3];
        float b2[3];
        {
            const float* bias = p.bias32 + ((size_t)amask * NH + h) * (3 * 576);
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                const int bl = qi < 2 ? lane : lane2;
                s[qi][0] = *(const float4v*)(bias + qi * 576 + bl * 4);
                s[qi][1] = *(const float4v*)(bias + qi * 576 + 256 + bl * 4);
                s[qi][2] = zero4;
                b2[qi] = bias[qi * 576 + 512 + bl];       // key tile 2 holds one key per lane: added after the product
            }
        }
        // ---- q^T, k^T (rows = features: A = weights, B = x) and v (rows = slab rows: A = x, B = weights)
        float4v aq[3] = {zero4, zero4, zero4}, ak[3] = {zero4, zero4, zero4}, av[3] = {zero4, zero4, zero4};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const half8 xf = *(const half8*)(Xs + (sbase + tt * 16 + fr) * LDX + ks * 32 + g * 8);
                aq[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[cur][ks], xf, aq[tt], 0, 0, 0);
                ak[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[cur][3 + ks], xf, ak[tt], 0, 0, 0);
                av[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, wr[cur][6 + ks], av[tt], 0, 0, 0);
            }
        }
        float4v bq = *(const float4v*)(p.bqkv + h * HD + g * 4);
        // The k bias adds the same q.bk to every key of a query and drops out of the softmax; the v bias commutes with the
        // weighted mean (sum p (v + bv) / sum p = sum p v / sum p + bv) and is added to the normalised output instead.
        const float4v bv = *(const float4v*)(p.bqkv + 2 * C + h * HD + g * 4);
        bq *= qscale;
        half4 qf[3], kf[3], vf[3];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int j = 0; j < 4; ++j) { qf[tt][j] = (_Float16)fmaf(aq[tt][j], qscale, bq[j]); kf[tt][j] = (_Float16)ak[tt][j]; vf[tt][j] = (_Float16)av[tt][j]; }
        // ---- S^T = K Q^T on top of the bias (k = the 16 features), softmax over the keys
#pragma unroll
        for (int qi = 0; qi < 3; ++qi)
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) s[qi][kt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[kt], qf[qi], s[qi][kt], 0, 0, 0);
        float mx[3];
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            s[qi][2][0] += b2[qi];
            float m = fmaxf(fmaxf(s[qi][0][0], s[qi][0][1]), fmaxf(s[qi][0][2], s[qi][0][3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) m = fmaxf(m, s[qi][1][j]);
            mx[qi] = fmaxf(m, s[qi][2][0]);
        }
        rows_max3(mx[0], mx[1], mx[2]);
        half4 pf[3][3];
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            const float2v m2 = {mx[qi], mx[qi]};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {   // the subtractions as v_pk_add_f32
                const float2v d0 = (float2v){s[qi][kt][0], s[qi][kt][1]} - m2, d1 = (float2v){s[qi][kt][2], s[qi][kt][3]} - m2;
                pf[qi][kt] = (half4){(_Float16)__builtin_amdgcn_exp2f(d0[0]), (_Float16)__builtin_amdgcn_exp2f(d0[1]),
                                     (_Float16)__builtin_amdgcn_exp2f(d1[0]), (_Float16)__builtin_amdgcn_exp2f(d1[1])};
            }
            pf[qi][2] = (half4){(_Float16)__builtin_amdgcn_exp2f(s[qi][2][0] - mx[qi]), (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};   // slab row 32 + 4g = token 32 + g
        }
        // ---- O^T = V^T P^T: rows = features, columns = queries (k = 16 keys per product); parked in Os (token order).
        // The softmax denominators come off the matrix pipe too: a ones matrix in place of V^T leaves the sum of the
        // (fp16) probabilities of query fr in every row of its column - the lane that scales the column already holds it.
        const half4 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
#pragma unroll
        for (int qi = 0; qi < 3; ++qi) {
            float4v o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[0], pf[qi][0], zero4, 0, 0, 0);
            float4v l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf[qi][0], zero4, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[1], pf[qi][1], o, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf[qi][1], l, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[2], pf[qi][2], o, 0, 0, 0);
            l = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pf[qi][2], l, 0, 0, 0);
            const float inv = __builtin_amdgcn_rcpf(l[0]);
            const int query = qi < 2 ? qi * 16 + fr : 32 + (fr >> 2);
            if (aok && (qi < 2 || (fr & 3) == 0)) {
                const float2v i2 = {inv, inv};
                const float2v o0 = __builtin_elementwise_fma((float2v){o[0], o[1]}, i2, (float2v){bv[0], bv[1]});
                const float2v o1 = __builtin_elementwise_fma((float2v){o[2], o[3]}, i2, (float2v){bv[2], bv[3]});
                const half4 oh = {(_Float16)o0[0], (_Float16)o0[1], (_Float16)o1[0], (_Float16)o1[1]};
                *(half4*)(Os + (tbase + query) * LDX + h * HD + g * 4) = oh;
            }
        }
    }
#undef W2X_LOAD_W
    __syncthreads();      // every wave's head outputs are in Os; nobody reads the slabs any more

    // the residual rows are fetched now, under the projection
    half8 xres[NPASS];
    int my_pix[NPASS];
    {
        const int li = tid & (LPR - 1);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            my_pix[ps] = Pix[ps * RPP + tid / LPR];
            half8 h = *(const half8*)(X + (size_t)(my_pix[ps] < 0 ? 0 : my_pix[ps]) * C + (li < PPR ? li : 0) * 8);
            if (!(my_pix[ps] >= 0 && li < PPR)) h = zero8;
            xres[ps] = h;
        }
    }
    // ---- proj: out = Os * Wproj^T + b -> tile over Xs.  10 units of (16-row tile, 3 n-tiles), weights as fragments from L2
    for (int u = wv; u < RT * 2; u += 4) {
        const int mt = u >> 1, n3 = (u & 1) * 3;
        half8 wf[3][3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wf[t][ks] = *(const half8*)(Wproj + (size_t)(((n3 + t) * 3 + ks) * 64 + lane) * 8);
        float4v acc[3] = {zero4, zero4, zero4};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const half8 of = *(const half8*)(Os + (mt * 16 + fr) * LDX + ks * 32 + g * 8);
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(of, wf[t][ks], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float b = p.bproj[(n3 + t) * 16 + fr];
#pragma unroll
            for (int j = 0; j < 4; ++j) Xs[(mt * 16 + g * 4 + j) * LDX + (n3 + t) * 16 + fr] = (_Float16)(acc[t][j] + b);
        }
    }
    __syncthreads();

    // ---- row pieces: + residual x, scatter store, LayerNorm statistics for the next op
    {
        _Float16* __restrict__ Y = (_Float16*)p.y;
        const int li = tid & (LPR - 1);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = ps * RPP + tid / LPR;
            const int pix = my_pix[ps];
            const bool ok = pix >= 0 && li < PPR;
            half8 o = {};
            if (ok) {
                o = *(const half8*)(Xs + r * LDX + li * 8) + xres[ps];
                *(half8*)(Y + (size_t)pix * C + li * 8) = o;
            }
            if (p.stats_out) {
                float s, q;
                sum_sq8(o, s, q);
                s = group_sum16(s);
                q = group_sum16(q);
                const float mean = s * (1.f / C);
                if (ok && li == 0) { p.stats_out[2 * (size_t)pix] = mean; p.stats_out[2 * (size_t)pix + 1] = rsqrtf(fmaxf(q * (1.f / C) - mean * mean, 0.f) + p.eps_out); }
            }
        }
    }
}

}  // namespace

hipError_t launch_swin_attn96_g2(const SwinAttnParams& p, hipStream_t s) {
    static unsigned lds_ok = 0;   // per-device bit: kernels.h ensure_dynamic_lds
    if (hipError_t e = ensure_dynamic_lds((const void*)swin_attn96_g2_kernel, SMEM96, lds_ok); e != hipSuccess) return e;
    const long total_win = (long)p.B * p.nwin;
    dim3 grid((unsigned)((total_win + G - 1) / G));
    hipLaunchKernelGGL(swin_attn96_g2_kernel, grid, dim3(256), SMEM96, s, p);
    return hipGetLastError();
}

}  // namespace w2x
